"""GPU parity of the decode-row kernels of the float weight types (round 4): nf4 / fp4 codes as int8 digit planes on the
int8 MFMA (csrc/woq_gemv_common.h LutArgs, inside woq_gemv_i8.hip / woq_gemv_xqs.h) and fp8 code bytes straight into the
fp8 MFMA (csrc/woq_gemv_fp8.hip), through qbits.woq_linear against the oracle (reference definition:
autograd/functions.py:41-63; weight strings bestla_weightonly_dispatcher.hpp:62-72). The quantise / dequantise / blob
parity of these types and their prefill-row path stay in test_gpu_parity.py; the fused engine with table-type layers in
test_gpu_engine.py.
"""
import numpy as np
import pytest
import torch

from oracle import woq_oracle as orc

pytestmark = pytest.mark.gpu

TABLE_TYPES = {"nf4": orc.W_NF4, "fp4_e2m1": orc.W_FP4_E2M1, "fp4_e2m1_bnb": orc.W_FP4_E2M1_BNB}
FP8_TYPES = {"fp8_e4m3": orc.W_FP8_E4M3, "fp8_e5m2": orc.W_FP8_E5M2}


@pytest.fixture(scope="module")
def qbits():
    from intel_extension_for_transformers_amd import qbits as q

    return q


@pytest.mark.parametrize("wname", sorted(TABLE_TYPES))
@pytest.mark.parametrize("K,N,group,sname", [(4096, 256, 128, "fp32"), (1024, 96, 32, "bf16"), (2048, 64, 64, "fp16"),
                                             (11008, 48, 128, "fp16"), (384, 272, -1, "fp32"), (160, 24, 64, "fp32")])
def test_table_weight_types_decode_kernel(qbits, wname, K, N, group, sname):
    """Round 4: nf4 / fp4 at decode row counts run the int4 MFMA kernel with a digit-plane unpack of the codes
    (csrc/woq_gemv_common.h LutArgs: table * S as one or three balanced int8 digits per code, one MFMA per plane) instead
    of the fp32 VALU kernel. 1..8 rows (one and two MFMA row sets), fp32 and 16-bit activation rows, every scale type,
    per-128 / per-32 / per-64 / per-channel groups, ragged N and K, K beyond one wave slice: within fp32 summation error
    of dequantise -> matmul -> + bias (nf4's table is held to 2^-23 of its largest entry; both fp4 tables exactly). The
    fp32 VALU kernel these types ran on before (still the path of misaligned rows and g_idx blobs; reached here through
    activation rows one element off 16-byte alignment) is held to the same bound."""
    wt = TABLE_TYPES[wname]
    rng = np.random.default_rng(43)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    q, s = orc.rtn_quantize_table(w, True, group, wt)
    st = {"fp32": orc.F32, "fp16": orc.F16, "bf16": orc.BF16}[sname]
    ref_blob = orc.repack_table(q, s, wt, group, scale_type=st)
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    want = orc.dequantize_blob(ref_blob)
    bias = rng.random(N, dtype=np.float32)
    # compute bf16: nf4 decodes with two digit planes — the table held to 2^-16 of its largest entry, 2e-4 of its
    # smallest (the reference's bf16 compute rounds every dequantised weight to 8 mantissa bits: 4e-3)
    for cname, rel in (("fp32", 2e-6), ("bf16", 2.5e-4 if wname == "nf4" else 2e-6)):
        blob = qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(), e8, e32, wname,
                                             sname, cname, False, group)
        if cname == "fp32":
            assert np.array_equal(blob.cpu().numpy().view(np.uint8), ref_blob)
        for M, adt in ((1, torch.float32), (2, torch.float32), (4, torch.float32), (5, torch.float32),
                       (8, torch.float32), (1, torch.float16), (3, torch.bfloat16)):
            x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(adt)
            xf = x.float().numpy()
            ref = orc.woq_linear(xf, ref_blob, bias)
            out = torch.full((M, N), float("nan"), device="cuda")
            qbits.woq_linear(x.cuda(), blob, torch.from_numpy(bias).cuda(), out, cname, wname, sname, False)
            mag = np.abs(xf) @ np.abs(want)
            assert (np.abs(out.cpu().numpy() - ref) <= rel * mag + 1e-5).all(), (cname, M, adt)
            if cname != "fp32":
                continue
            # the same call with the activation rows one element off 16-byte alignment: the generic fp32 kernel
            xp = torch.zeros(M, K + 8, dtype=adt, device="cuda")[:, 1:K + 1]
            xp.copy_(x)
            out2 = torch.full((M, N), float("nan"), device="cuda")
            qbits.woq_linear(xp, blob, torch.from_numpy(bias).cuda(), out2, "fp32", wname, sname, False)
            assert (np.abs(out2.cpu().numpy() - ref) <= 2e-6 * mag + 1e-5).all(), (M, adt, "generic")


@pytest.mark.parametrize("wname,sname", [("fp8_e4m3", "fp32"), ("fp8_e4m3", "fp8_e8m0"), ("fp8_e5m2", "fp32"),
                                         ("fp8_e5m2", "fp8_e8m0")])
@pytest.mark.parametrize("K,N,group", [(4096, 128, 128), (384, 64, -1), (11008, 64, 128), (8320, 32, 128),
                                       (4096, 64, 32), (1024, 48, 64), (11008, 32, 32), (768, 32, 96)])
def test_fp8_weight_types_decode_kernel(qbits, wname, sname, K, N, group):
    """Round 4: fp8 weights at decode row counts on the fp8 matrix cores (csrc/woq_gemv_fp8.hip): the code bytes are the
    B operand of v_mfma_f32_16x16x32_fp8_{fp8,bf8} as they are, the fp32 activation goes in as six balanced base-16
    digits (exact e4m3 values), fp32 recombination. 1..8 rows (one and two rows per MFMA row set, up to four sets),
    fp32 / fp16 / bf16 rows, fp32 and power-of-two scales, per-128 groups and one group per column; K = 11008 (a 7B
    down_proj: eleven waves of eight tiles — round 5's form for K > 8192) and K = 8320 (its shortest case: nine waves,
    the last with one tile); groups of 32 (the reference's default group size), 64 and 96 (round 5: one scale per 32-k
    block, each 64-k half issued once per block with the other block's A rows read as zeros). Bound: 1e-5 of
    sum |x||w| + 1e-5 — the matrix core's accumulation is not an fp32 adder (it aligns a dot's 32 products to the largest
    one): measured 2e-6 on RTN weights like these, 4.6e-6 on a matrix holding every finite e4m3 code at full scale
    (profiles/r04ah_fp8_decode.txt); test_fp8_weight_types_quantize_dequant_linear holds the same kernel to 2e-6 at its
    shapes. Second pass: a matrix of EVERY finite code (subnormals, both zeros) at 2^-10 of the scale — no code may
    produce garbage; the absolute term carries that comparison."""
    wt, e8 = FP8_TYPES[wname], sname == "fp8_e8m0"
    rng = np.random.default_rng(63)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    q, s = orc.rtn_quantize_fp8(w, True, group, wt, e8)
    finite = np.flatnonzero(np.isfinite(orc.FP8_TABLES[wt])).astype(np.uint8)
    q_all = finite[rng.integers(0, finite.size, size=(K, N))]
    e8t, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    bias = rng.random(N, dtype=np.float32)
    for codes, sc in ((q, s), (q_all, (s * np.float32(2.0 ** -10)).astype(np.float32))):
        ref_blob = orc.repack_fp8(codes, sc, wt, None, group, e8m0=e8)
        blob = qbits.repack_quantized_weight(torch.from_numpy(codes.view(np.int8)).cuda(), torch.from_numpy(sc).cuda(),
                                             e8t, e32, wname, sname, "fp32", False, group)
        assert np.array_equal(blob.cpu().numpy().view(np.uint8), ref_blob)
        want = orc.dequantize_blob(ref_blob)
        for M, adt in ((1, torch.float32), (2, torch.float32), (3, torch.float32), (8, torch.float32),
                       (1, torch.float16), (5, torch.bfloat16)):
            x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(adt)
            xf = x.float().numpy()
            ref = orc.woq_linear(xf, ref_blob, bias)
            out = torch.full((M, N), float("nan"), device="cuda")
            qbits.woq_linear(x.cuda(), blob, torch.from_numpy(bias).cuda(), out, "fp32", wname, sname, False)
            mag = np.abs(xf) @ np.abs(want)
            assert (np.abs(out.cpu().numpy() - ref) <= 1e-5 * mag + 1e-5).all(), (M, adt)
            # rows one element off 16-byte alignment: the lookup kernel (fp32 arithmetic)
            xp = torch.zeros(M, K + 8, dtype=adt, device="cuda")[:, 1:K + 1]
            xp.copy_(x)
            out2 = torch.full((M, N), float("nan"), device="cuda")
            qbits.woq_linear(xp, blob, torch.from_numpy(bias).cuda(), out2, "fp32", wname, sname, False)
            assert (np.abs(out2.cpu().numpy() - ref) <= 2e-6 * mag + 1e-5).all(), (M, adt, "lookup kernel")
