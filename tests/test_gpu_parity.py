"""GPU parity: the HIP path (through the C ABI / qbits shim) against the CPU oracle on identical inputs.

Bit-exact for bytes / integers / indices; floating point with the tolerance written at each assert.
Shapes the oracle finishes in seconds; full-size properties are in test_gpu_fullsize.py.
"""
import numpy as np
import pytest
import torch

from oracle import woq_oracle as orc

pytestmark = pytest.mark.gpu

DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
ST = {"fp32": orc.F32, "bf16": orc.BF16, "fp16": orc.F16}


@pytest.fixture(scope="module")
def qbits():
    from intel_extension_for_transformers_amd import qbits as q

    return q


def _mk(K, N, group, asym, shuf, seed=0):
    rng = np.random.default_rng(seed)
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    g = K if group == -1 else group
    G = (K + g - 1) // g
    s = (rng.random((G, N), dtype=np.float32) + 0.5) * 0.01
    z = rng.integers(-8, 8, (G, N), dtype=np.int8) if asym else None
    # raw GPTQ g_idx: group id of every K row, each group exactly g rows, in act-order (shuffled) positions —
    # what the reference hands to repack_quantized_weight (qbits_ut/test_packq.py:59-64)
    idx = rng.permutation(np.arange(K, dtype=np.int32) // g).astype(np.int32) if shuf else None
    return q, s, z, idx


def _cvt(idx, K, group):
    """raw g_idx -> activation shuffle indices (the reference's convert_idx, test_packq.py:22-28), for the oracle."""
    return None if idx is None else orc.convert_idx(idx, K, K if group == -1 else group)


def _gpu_blob(qbits, q, s, z, idx, group, scale_type="fp32"):
    e8 = torch.empty(0, dtype=torch.int8)
    e32 = torch.empty(0, dtype=torch.int32)
    return qbits.repack_quantized_weight(
        torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(), e8 if z is None else torch.from_numpy(z).cuda(),
        e32 if idx is None else torch.from_numpy(idx).cuda(), "int4_clip", scale_type, "fp32", z is not None, group)


CASES = [
    (512, 1024, 128, False, False),  # qbits_ut/test_weightonly.py geometry
    (512, 1024, 128, True, True),    # qbits_ut/test_packq.py geometry
    (512, 1024, -1, False, False),
    (256, 48, 32, True, False),      # scale_mode 1
    (160, 24, 64, True, False),      # tail group, K/N padding
    (96, 20, 32, False, False),
    (16, 8, -1, False, False),
    (1024, 4096, 128, False, False),
]


@pytest.mark.parametrize("K,N,group,asym,shuf", CASES)
@pytest.mark.parametrize("scale_type", ["fp32", "fp16", "bf16"])
def test_repack_blob_bit_exact(qbits, K, N, group, asym, shuf, scale_type):
    """Device repack == oracle repack, byte for byte (layout transform only, qbits.cpp:61-77)."""
    q, s, z, idx = _mk(K, N, group, asym, shuf)
    blob = _gpu_blob(qbits, q, s, z, idx, group, scale_type)
    ref = orc.repack(q, s, z, _cvt(idx, K, group), group, scale_type=ST[scale_type])
    got = blob.cpu().numpy().view(np.uint8)
    assert got.size == ref.size == qbits.get_packed_weight_size(K, N, "int4_clip", scale_type, "fp32", asym, group,
                                                                shuf)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("K,N,group,asym,shuf", CASES)
def test_dequantize_exact(qbits, K, N, group, asym, shuf):
    """(q - zp) * scale is one fp32 multiply: exact against the oracle, both orientations (qbits.cpp:102-111)."""
    q, s, z, idx = _mk(K, N, group, asym, shuf, seed=1)
    blob = _gpu_blob(qbits, q, s, z, idx, group)
    ref = orc.dequant_raw(q, s, z, K if group == -1 else group)
    out = torch.zeros(K, N, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, out, False, "fp32", "int4_clip", "fp32")
    assert np.array_equal(out.cpu().numpy(), ref)
    out_t = torch.zeros(N, K, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, out_t, True, "fp32", "int4_clip", "fp32")
    assert np.array_equal(out_t.cpu().numpy(), ref.T)


def test_packed_weight_info_roundtrip(qbits):
    """qbits_ut/test_packq.py:88-109: size, type strings, act-shuffle flag, g_idx / scale / zp pass-through exact."""
    K, N, bs = 512, 1024, 128
    torch.manual_seed(0)
    raw = torch.randint(-8, 8, [K, N], dtype=torch.int8)
    g_idx = torch.arange(K // bs, dtype=torch.int).repeat(bs)
    cvt = torch.from_numpy(orc.convert_idx(g_idx.numpy(), K, bs))
    zp = torch.randint(-4, 4, [K // bs, N], dtype=torch.int8)
    scale = torch.rand(K // bs, N, dtype=torch.float)
    packw = qbits.repack_quantized_weight(raw.cuda(), scale.cuda(), zp.cuda(), g_idx.cuda(), "int4_clip", "fp32",
                                          "fp32", True, bs)
    assert qbits.acquire_packed_weight_info(packw, 0)[0].item() == packw.numel()
    assert qbits.acquire_packed_weight_info(packw, 1)[0].item() == bs
    assert qbits.acquire_packed_weight_info(packw, 2)[0].item() == K
    assert qbits.acquire_packed_weight_info(packw, 3)[0].item() == N
    assert qbits.acquire_packed_weight_info(packw, 4)[0].item() == 1
    assert qbits.acquire_packed_weight_info(packw, 11)[0].item() == 1
    name = "".join(chr(c) for c in qbits.acquire_packed_weight_info(packw, 6).tolist())
    assert name == "int4_clip"
    assert "".join(chr(c) for c in qbits.acquire_packed_weight_info(packw, 7).tolist()) == "fp32"
    assert "".join(chr(c) for c in qbits.acquire_packed_weight_info(packw, 8).tolist()) == "fp32"
    assert torch.equal(qbits.acquire_packed_weight_info(packw, 5).cpu(), cvt.to(torch.int32))
    assert torch.equal(qbits.acquire_packed_weight_info(packw, 9).cpu(), scale)
    assert torch.equal(qbits.acquire_packed_weight_info(packw, 10).cpu(), zp)


@pytest.mark.parametrize("K,N,group,asym,shuf", CASES)
@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 16])
def test_woq_linear_decode_vs_oracle(qbits, K, N, group, asym, shuf, M):
    """Decode GEMV (M <= 8) vs the parity definition (autograd/functions.py:41-63), fp32 in / fp32 out.
    Tolerance: |err| <= 2e-5 * sum_k |x_k w_k| bound, stated as 1e-4 * max|ref| + 1e-6 (fp32 accumulation in a
    different order than the oracle's double sum); the reference's own criterion is allclose(rtol=0.03)."""
    q, s, z, idx = _mk(K, N, group, asym, shuf, seed=2)
    blob = _gpu_blob(qbits, q, s, z, idx, group)
    rng = np.random.default_rng(3)
    x = (rng.random((M, K), dtype=np.float32) - 0.3)
    bias = rng.random(N, dtype=np.float32) * 10
    ref = orc.woq_linear(x, orc.repack(q, s, z, _cvt(idx, K, group), group), bias)
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    qbits.woq_linear(torch.from_numpy(x).cuda(), blob, torch.from_numpy(bias).cuda(), out, "fp32", "int4_clip", "fp32",
                     asym)
    got = out.cpu().numpy()
    tol = 1e-4 * np.abs(ref - bias).max() + 1e-6
    assert np.abs(got - ref).max() <= tol
    assert np.allclose(got, ref, rtol=0.03, atol=tol)


@pytest.mark.parametrize("K,N,group,asym", [(4096, 4096, 128, False), (11008, 4096, 128, True), (4096, 256, 32, True),
                                            (4096, 512, 32, False), (640, 96, 64, True), (1536, 48, -1, False)])
@pytest.mark.parametrize("scale_type,src_dt", [("fp32", "fp32"), ("fp16", "fp32"), ("bf16", "fp16"), ("fp16", "bf16")])
def test_act_order_decode_gather_vs_oracle(qbits, monkeypatch, K, N, group, asym, scale_type, src_dt):
    """GPTQ act-order blobs at batch 1 (round 5): the i8-MFMA tile GEMV gathers its activations by the stored shuffle
    (`index_select(x, 1, g_idx)` of the parity definition, autograd/functions.py:41-63; BesTLA's shuffle prologue,
    bestla_weightonly_dispatcher.cpp:138-142) instead of handing such layers to the generic fp32 kernel. 7B projection
    shapes, per-128 / per-32 / per-64 / per-channel groups, a K with a tail slice, 16-bit rows, every scale type — vs the
    oracle at the decode tolerance, and equal (to that tolerance) to the generic kernel's answer for the same call."""
    q, s, z, idx = _mk(K, N, group, asym, True, seed=21)
    blob = _gpu_blob(qbits, q, s, z, idx, group, scale_type)
    xt = (torch.rand(1, K) - 0.3).to(DT[src_dt])
    ref = orc.woq_linear(xt.float().numpy(), orc.repack(q, s, z, _cvt(idx, K, group), group, scale_type=ST[scale_type]))
    out = torch.zeros(1, N, dtype=torch.float32, device="cuda")
    qbits.woq_linear(xt.cuda(), blob, torch.empty(0), out, "fp32", "int4_clip", scale_type, asym)
    got = out.cpu().numpy()
    tol = 1e-4 * np.abs(ref).max() + 1e-6
    assert np.abs(got - ref).max() <= tol
    out4 = torch.zeros(4, N, dtype=torch.float32, device="cuda")  # several rows: the generic kernel, same definition
    qbits.woq_linear(xt.repeat(4, 1).cuda(), blob, torch.empty(0), out4, "fp32", "int4_clip", scale_type, asym)
    assert np.abs(out4.cpu().numpy() - ref).max() <= tol


@pytest.mark.parametrize("src_dt,dst_dt", [("bf16", "bf16"), ("fp16", "fp16"), ("bf16", "fp32"), ("fp32", "bf16")])
@pytest.mark.parametrize("M", [1, 4])
def test_woq_linear_decode_dtypes(qbits, src_dt, dst_dt, M):
    """src/dst dtype matrix of qbits_ut/test_weightonly.py:40-41 (+fp16). The 16-bit activation is up-cast exactly
    to fp32 (as modules.py:153 does), so the only extra error is the final store rounding: bf16 2^-8, fp16 2^-11
    relative, on top of the fp32 tolerance."""
    K, N, group = 512, 1024, 128
    q, s, z, idx = _mk(K, N, group, True, False, seed=4)
    blob = _gpu_blob(qbits, q, s, z, idx, group)
    xt = (torch.rand(M, K) - 0.3).to(DT[src_dt])
    x = xt.float().numpy()
    ref = orc.woq_linear(x, orc.repack(q, s, z, _cvt(idx, K, group), group))
    out = torch.zeros(M, N, dtype=DT[dst_dt], device="cuda")
    qbits.woq_linear(xt.cuda(), blob, torch.empty(0), out, "fp32", "int4_clip", "fp32", True)
    got = out.float().cpu().numpy()
    store_eps = {"fp32": 0.0, "bf16": 2.0 ** -8, "fp16": 2.0 ** -11}[dst_dt]
    tol = (1e-4 + store_eps) * np.abs(ref).max() + 1e-6
    assert np.abs(got - ref).max() <= tol


@pytest.mark.parametrize("src_dt", ["bf16", "fp16"])
@pytest.mark.parametrize("K,group", [(640, 128), (11008, 128), (4096, 32)])
def test_woq_linear_16bit_rows_read_natively(qbits, src_dt, K, group):
    """16-bit activation rows go to the tile GEMV as they are (8-byte loads widened in registers): a K that is not a
    multiple of the 1024-element staging step, a K that is (several slices), a strided view (lda > K), M up to 7
    (two row chunks). Same bound as fp32 rows — the widening is exact."""
    N, M = 256, 7
    q, s, z, idx = _mk(K, N, group, True, False, seed=11)
    blob = _gpu_blob(qbits, q, s, z, idx, group)
    wide = (torch.rand(M, K + 64) - 0.4).to(DT[src_dt])
    xt = wide[:, :K]
    ref = orc.woq_linear(np.ascontiguousarray(xt.float().numpy()), orc.repack(q, s, z, _cvt(idx, K, group), group))
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    qbits.woq_linear(wide.cuda()[:, :K], blob, torch.empty(0), out, "fp32", "int4_clip", "fp32", True)
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("blocksize,asym", [(128, False), (128, True), (-1, False), (32, True)])
@pytest.mark.parametrize("add_bias", [True, False])
def test_reference_unit_test_idiom(qbits, blocksize, asym, add_bias):
    """qbits_ut/test_weightonly.py:51-88 run against the HIP library: seed 0, uniform [0,1) inputs, m=256 n=1024
    k=512, quantize_to_packed_weight -> dequantize_packed_weight -> torch.matmul reference,
    torch.allclose(rtol=0.03), both transpose settings, fp32 and bf16 src/dst."""
    m, n, k = 256, 1024, 512
    for transpose in (True, False):
        for src_dt, dst_dt in (("fp32", "fp32"), ("bf16", "bf16")):
            torch.manual_seed(0)
            ref_activation = torch.rand(m, k, dtype=torch.float)
            tar_activation = ref_activation.clone().to(DT[src_dt])
            wei_row, wei_col = (n, k) if transpose else (k, n)
            raw_wei = torch.rand(wei_row, wei_col, dtype=torch.float)
            compress_wei = qbits.quantize_to_packed_weight(raw_wei.cuda(), transpose, blocksize, "fp32", "int4_clip",
                                                           "fp32", asym)
            revert_wei = torch.zeros(wei_row, wei_col, dtype=torch.float, device="cuda")
            qbits.dequantize_packed_weight(compress_wei, revert_wei, transpose, "fp32", "int4_clip", "fp32")
            revert_wei = revert_wei.cpu()
            bias = torch.rand(n, dtype=torch.float) * 10 if add_bias else torch.empty(0)
            tar_dst = torch.zeros(m, n, dtype=DT[dst_dt], device="cuda")
            if transpose:
                revert_wei = torch.transpose(revert_wei, 0, 1)
            ref_dst = torch.matmul(ref_activation, revert_wei)
            qbits.woq_linear(tar_activation.cuda(), compress_wei, bias.cuda(), tar_dst, "fp32", "int4_clip", "fp32",
                             asym)
            tar = tar_dst.float().cpu()
            if add_bias:
                ref_dst += bias
            assert torch.allclose(tar, ref_dst, rtol=0.03), (transpose, src_dt)


def test_rtn_quantizer_matches_oracle(qbits):
    """Device RTN == oracle RTN bit for bit on the blob (same formula, PARITY UNPINNED vs BesTLA — see DESIGN.md)."""
    torch.manual_seed(5)
    w = torch.randn(256, 96)
    for transpose in (False, True):
        for group, asym in ((32, False), (128, True), (-1, False)):
            wt = w.t().contiguous() if transpose else w
            blob = qbits.quantize_to_packed_weight(wt.cuda(), transpose, group, "fp32", "int4_clip", "fp32", asym)
            q, s, z = orc.rtn_quantize(wt.numpy(), transpose, group, asym)
            ref = orc.repack(q, s, z, None, group)
            assert np.array_equal(blob.cpu().numpy().view(np.uint8), ref)


def test_ops_vs_oracle_and_hf_golden(qbits, golden_dir):
    """RMSNorm / RoPE / SiLU*mul / GeLU kernels vs HF outputs (tests/golden/hf_ops.npz). fp32 tolerance 1e-5."""
    import os

    g = np.load(os.path.join(golden_dir, "hf_ops.npz"))
    x = torch.from_numpy(g["rms_x"]).cuda()
    y = qbits.rmsnorm(x, torch.from_numpy(g["rms_w"]).cuda(), float(g["rms_eps"]))
    np.testing.assert_allclose(y.cpu().numpy(), g["rms_y"], rtol=1e-5, atol=1e-5)
    from intel_extension_for_transformers_amd.runtime import build_rope_tables

    q = torch.from_numpy(g["rope_q"][0]).permute(1, 0, 2).contiguous().cuda()  # [tokens, heads, D]
    pos = torch.from_numpy(g["rope_pos"][0].astype(np.int32)).cuda()
    cos, sin = build_rope_tables(8192, q.shape[-1], 10000.0, "cuda")
    qbits.rope(q, pos, cos, sin)
    np.testing.assert_allclose(q.permute(1, 0, 2).cpu().numpy(), g["rope_qr"][0], rtol=0, atol=2e-5)
    gate, up = torch.from_numpy(g["silu_gate"]).cuda(), torch.from_numpy(g["silu_up"]).cuda()
    np.testing.assert_allclose(qbits.silu_mul(gate, up).cpu().numpy(), g["silu_y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(qbits.gelu(gate, "tanh").cpu().numpy(), g["gelu_new_y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(qbits.gelu(gate, "erf").cpu().numpy(), g["gelu_y"], rtol=1e-5, atol=1e-5)
    # 16-bit storage variants: compute in fp32, round once on store
    yb = qbits.rmsnorm(x.bfloat16(), torch.from_numpy(g["rms_w"]).cuda(), float(g["rms_eps"]))
    ref = orc.rmsnorm(x.bfloat16().float().cpu().numpy(), g["rms_w"], float(g["rms_eps"]))
    np.testing.assert_allclose(yb.float().cpu().numpy(), ref, rtol=2.0 ** -7, atol=1e-3)


def test_errors_are_runtime_errors(qbits):
    """Error convention: RuntimeError with a 'QBits:' / 'Qbits:' prefix (dispatcher.cpp:289,368; qbits.cpp:35)."""
    with pytest.raises(RuntimeError, match="[Qq][Bb]its"):
        qbits.repack_quantized_weight(torch.zeros(96, 16, dtype=torch.int8).cuda(), torch.ones(6, 16).cuda(),
                                      torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32), "int4_clip",
                                      "fp32", "fp32", False, 16)
    blob = qbits.quantize_to_packed_weight(torch.rand(64, 32).cuda(), False, 32, "fp32", "int4_clip", "fp32", False)
    with pytest.raises(RuntimeError, match="QBits"):
        qbits.woq_linear(torch.rand(1, 65).cuda(), blob, torch.empty(0), torch.zeros(1, 32).cuda(), "fp32",
                         "int4_clip", "fp32", False)
    with pytest.raises(RuntimeError, match="unsupported qbits data type"):
        qbits.woq_linear(torch.rand(1, 64).double().cuda(), blob, torch.empty(0), torch.zeros(1, 32).cuda(), "fp32",
                         "int4_clip", "fp32", False)
    with pytest.raises(RuntimeError, match="[Qq]bits"):
        qbits.quantize_to_packed_weight(torch.rand(64, 32).cuda(), False, 32, "fp32", "int5_clip", "fp32", False)


@pytest.mark.parametrize("K,N,group,asym", [(512, 1024, 128, False), (512, 1024, 128, True), (256, 48, 32, True),
                                            (160, 24, 64, True), (384, 200, -1, False), (1024, 4096, 256, False),
                                            (2048, 520, 32, True)])
@pytest.mark.parametrize("M", [9, 40, 130, 256])
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_woq_linear_prefill_gemm_vs_oracle(qbits, K, N, group, asym, M, compute):
    """MFMA GEMM path (M > 16, csrc/woq_gemm_f16.hip; 9 rows: the decode GEMV's third row set) vs the parity definition,
    ragged M / N / K included. The K >= 1024 shapes at these row counts launch few workgroups and run as K slices
    (split-K: scaled fp32 partials per slice, summed in a fixed order by splitk_reduce_kernel with the bias).
    compute_dtype fp32: activations and scaled weights as hi + lo fp16 pairs (~22 bits each), three products per
    pair -> fp32-class, stated bound 2e-5 * sum|x||w| expressed as 1e-4 * max|ref| + 1e-5. compute_dtype bf16: one plane,
    activation rounded to fp16 (2^-11 relative): stated bound 2e-3 * max|ref| (the reference's own criterion for
    reduced-precision compute is allclose(rtol=0.03), qbits_ut/test_weightonly.py:88)."""
    q, s, z, idx = _mk(K, N, group, asym, False, seed=7)
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    blob = qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                         e8 if z is None else torch.from_numpy(z).cuda(), e32, "int4_clip", "fp32",
                                         compute, z is not None, group)
    rng = np.random.default_rng(8)
    x = (rng.random((M, K), dtype=np.float32) - 0.3) * np.exp(rng.normal(0, 2, (M, 1))).astype(np.float32)
    bias = rng.random(N, dtype=np.float32)
    ref = orc.woq_linear(x, orc.repack(q, s, z, None, group), bias)
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    qbits.woq_linear(torch.from_numpy(x).cuda(), blob, torch.from_numpy(bias).cuda(), out, compute, "int4_clip",
                     "fp32", asym)
    got = out.cpu().numpy()
    # per-row bound: rows differ in magnitude by e^2-ish factors on purpose (block floating point is per row)
    scale = np.abs(ref - bias).max(axis=1, keepdims=True)
    rel = 1e-4 if compute == "fp32" else 2e-3
    assert (np.abs(got - ref) <= rel * scale + 1e-5).all()


@pytest.mark.parametrize("act_dt,out_dt,scale_type", [("bf16", "bf16", "fp16"), ("fp16", "fp16", "bf16"),
                                                      ("fp32", "bf16", "fp32"), ("bf16", "fp32", "fp16")])
@pytest.mark.parametrize("K,N,group,asym,shuf", [(512, 1024, 128, False, False), (512, 1024, 128, True, True),
                                                 (256, 48, 32, True, False), (160, 24, 64, True, False),
                                                 (2048, 272, 32, False, True)])
def test_woq_linear_prefill_f16_operand_variants(qbits, K, N, group, asym, shuf, act_dt, out_dt, scale_type):
    """fp16-operand MFMA GEMM (compute_dtype bf16, csrc/woq_gemm_f16.hip): activation / output / scale dtypes, the
    GPTQ act-order gather done in the pack pass, K / N / M tails. Bound: operands carry 11-bit significands
    (2^-12 relative per product, the reference's bf16 cores 2^-9) -> 2e-3 * rowmax|ref| as in the test above, plus
    the output type's own rounding (bf16 2^-9, fp16 2^-11 of the value)."""
    M = 200
    q, s, z, idx = _mk(K, N, group, asym, shuf, seed=11)
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    blob = qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                         e8 if z is None else torch.from_numpy(z).cuda(),
                                         e32 if idx is None else torch.from_numpy(idx).cuda(), "int4_clip", scale_type,
                                         "bf16", z is not None, group)
    rng = np.random.default_rng(12)
    x32 = (rng.random((M, K), dtype=np.float32) - 0.4) * np.exp(rng.normal(0, 1.5, (M, 1))).astype(np.float32)
    xt = torch.from_numpy(x32).to(DT[act_dt])
    x_used = xt.float().numpy()  # what the kernel is given, exactly
    s_used = torch.from_numpy(s).to(DT[scale_type]).float().numpy()
    ref = orc.woq_linear(x_used, orc.repack(q, s_used, z, _cvt(idx, K, group), group), None)
    out = torch.zeros(M, N, dtype=DT[out_dt], device="cuda")
    qbits.woq_linear(xt.cuda(), blob, torch.empty(0), out, "bf16", "int4_clip", scale_type, asym)
    got = out.float().cpu().numpy()
    scale = np.abs(ref).max(axis=1, keepdims=True)
    out_eps = {"fp32": 0.0, "bf16": 2.0 ** -8, "fp16": 2.0 ** -10}[out_dt]
    assert (np.abs(got - ref) <= 2e-3 * scale + out_eps * np.abs(ref) + 1e-5).all()


@pytest.mark.parametrize("group,asym,scale_type", [(128, False, "fp16"), (32, True, "fp16"), (32, True, "fp32"),
                                                   (128, True, "bf16"), (64, False, "fp32")])
@pytest.mark.parametrize("M,K,N", [(9, 4096, 1152), (130, 1024, 1152), (130, 640, 256)])
def test_woq_linear_prefill_stores_every_element_and_repeats(qbits, M, K, N, group, asym, scale_type):
    """The hand-scheduled K loop (csrc/woq_gemm_f16p.h; K = 640 has an odd tile count and stays on hipcc's schedule):
    the output is pre-filled with NaN before every launch, so an element a workgroup did not store is seen (that
    happened on cold starts until the epilogue's arguments were pinned in SGPRs), every repeat is bit-identical, and
    the values are the oracle's within the fp16-operand bound."""
    q, s, z, idx = _mk(K, N, group, asym, False, seed=21)
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    blob = qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                         e8 if z is None else torch.from_numpy(z).cuda(), e32, "int4_clip", scale_type,
                                         "bf16", z is not None, group)
    rng = np.random.default_rng(22)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).cuda()
    s_used = torch.from_numpy(s).to(DT[scale_type]).float().numpy()
    ref = orc.woq_linear(x.cpu().numpy(), orc.repack(q, s_used, z, None, group), None)
    first = None
    for _ in range(12):
        out = torch.full((M, N), float("nan"), device="cuda")
        qbits.woq_linear(x, blob, torch.empty(0), out, "bf16", "int4_clip", scale_type, asym)
        assert not torch.isnan(out).any(), "an output element was never stored"
        if first is None:
            first = out
        else:
            assert torch.equal(out, first)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(first.cpu().numpy() - ref) <= 2e-3 * scale + 1e-5).all()


def test_woq_linear_prefill_f16_strided_rows(qbits):
    """lda / ldo larger than K / N and not 16-byte multiples: the generic pack path and the scalar-store epilogue."""
    K, N, M, group = 384, 100, 70, 128
    q, s, z, idx = _mk(K, N, group, False, False, seed=13)
    blob = qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                         torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32),
                                         "int4_clip", "fp32", "bf16", False, group)
    rng = np.random.default_rng(14)
    xbig = torch.from_numpy(rng.standard_normal((M, K + 3), dtype=np.float32)).cuda()
    obig = torch.full((M, N + 1), 7.0, device="cuda")
    x, out = xbig[:, :K], obig[:, :N]
    import ctypes

    from intel_extension_for_transformers_amd import _lib as L

    hdr = qbits.header_of(blob)
    L.check(L.lib().woq_linear(ctypes.c_void_p(x.data_ptr()), L.torch_dtype_code(x.dtype), x.stride(0),
                               ctypes.c_void_p(blob.data_ptr()), ctypes.byref(hdr), None,
                               ctypes.c_void_p(out.data_ptr()), L.torch_dtype_code(out.dtype), out.stride(0), M,
                               L.stream_ptr()))
    ref = orc.woq_linear(x.cpu().numpy().copy(), orc.repack(q, s, None, None, group), None)
    got = out.cpu().numpy()
    assert (np.abs(got - ref) <= 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 1e-5).all()
    assert (obig[:, N].cpu().numpy() == 7.0).all()  # the column past N is untouched


# ---- int8 weights (reference weight type "int8", bits = 8): composite of two int4 blobs ---------------------------
def _mk8(K, N, group, asym, shuf, seed=0):
    rng = np.random.default_rng(seed)
    q = rng.integers(-128, 128, (K, N), dtype=np.int8)
    g = K if group == -1 else group
    G = (K + g - 1) // g
    s = (rng.random((G, N), dtype=np.float32) + 0.5) * 0.001
    z = rng.integers(-128, 128, (G, N), dtype=np.int8) if asym else None
    idx = rng.permutation(np.arange(K, dtype=np.int32) // g).astype(np.int32) if shuf else None
    return q, s, z, idx


def _gpu_blob8(qbits, q, s, z, idx, group, scale_type="fp32", compute="fp32"):
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    return qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                         e8 if z is None else torch.from_numpy(z).cuda(),
                                         e32 if idx is None else torch.from_numpy(idx).cuda(), "int8", scale_type,
                                         compute, z is not None, group)


INT8_CASES = [(512, 1024, 128, False, False), (512, 1024, 128, True, True), (256, 48, 32, True, False),
              (160, 24, 64, False, False), (512, 64, -1, True, False)]


@pytest.mark.parametrize("K,N,group,asym,shuf", INT8_CASES)
@pytest.mark.parametrize("scale_type", ["fp32", "bf16"])
def test_int8_blob_bytes_dequant_and_info(qbits, K, N, group, asym, shuf, scale_type):
    """The int8 composite blob is byte-identical to the oracle's (split q8 / zp8 / scale exactly, then two int4
    repacks); dequantisation equals the oracle's two-term fp32 sum bit for bit and the one-multiply definition
    (q8 - zp8) * s to 2 ulp; acquire_packed_weight_info hands the original scale / zero-point tensors back."""
    q, s, z, idx = _mk8(K, N, group, asym, shuf, seed=21)
    blob = _gpu_blob8(qbits, q, s, z, idx, group, scale_type)
    ref = orc.repack_int8(q, s, z, _cvt(idx, K, group), group, scale_type=ST[scale_type])
    got = blob.cpu().numpy().view(np.uint8)
    assert got.size == ref.size == qbits.get_packed_weight_size(K, N, "int8", scale_type, "fp32", asym, group, shuf)
    assert np.array_equal(got, ref)
    deq = torch.empty(K, N, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, deq, False, "fp32", "int8", scale_type)
    want = orc.dequantize_blob(ref)
    assert np.array_equal(deq.cpu().numpy(), want)
    g = K if group == -1 else group
    s_used = torch.from_numpy(s).to(DT[scale_type]).float().numpy()
    zz = np.zeros_like(s_used) if z is None else z.astype(np.float32)
    plain = (q.astype(np.float32) - np.repeat(zz, g, 0)[:K]) * np.repeat(s_used, g, 0)[:K]
    assert np.abs(want - plain).max() <= 2.5e-7 * np.abs(plain).max()
    assert "".join(chr(c) for c in qbits.acquire_packed_weight_info(blob, 6).tolist()) == "int8"
    assert np.array_equal(qbits.acquire_packed_weight_info(blob, 9).cpu().numpy(), s_used)
    if asym:
        assert np.array_equal(qbits.acquire_packed_weight_info(blob, 10).cpu().numpy(), z)


@pytest.mark.parametrize("K,N,group,asym,shuf", INT8_CASES)
@pytest.mark.parametrize("M,compute", [(1, "fp32"), (4, "fp32"), (40, "fp32"), (200, "bf16")])
def test_int8_woq_linear_vs_oracle(qbits, K, N, group, asym, shuf, M, compute):
    """woq_linear on int8 weights = two int4 launches (HI into an fp32 scratch, LO adds it + bias): decode GEMV,
    fp32-compute GEMM and fp16-operand GEMM routes against dequantise -> matmul -> + bias of the oracle. Bounds as for
    the int4 routes they reuse (fp32-class 1e-4 of the row's |x||w| scale; 2e-3 for the reduced-precision mode)."""
    q, s, z, idx = _mk8(K, N, group, asym, shuf, seed=22)
    blob = _gpu_blob8(qbits, q, s, z, idx, group, "fp32", compute)
    rng = np.random.default_rng(23)
    x = rng.standard_normal((M, K)).astype(np.float32)
    bias = rng.random(N, dtype=np.float32)
    ref = orc.woq_linear(x, orc.repack_int8(q, s, z, _cvt(idx, K, group), group), bias)
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    qbits.woq_linear(torch.from_numpy(x).cuda(), blob, torch.from_numpy(bias).cuda(), out, compute, "int8", "fp32", asym)
    got = out.cpu().numpy()
    # cancellation-aware scale: sum |x| |w| per row/column is what the error is relative to
    w = np.abs(orc.dequantize_blob(orc.repack_int8(q, s, z, None, group)))
    mag = np.abs(x) @ w if idx is None else np.abs(x)[:, _cvt(idx, K, group)] @ w
    rel = 2e-6 if compute == "fp32" else 2e-3
    assert (np.abs(got - ref) <= rel * mag + 1e-5).all()


def test_int8_quantize_to_packed_weight_roundtrip(qbits):
    """qbits.quantize_to_packed_weight(weight_type="int8") == the 8-bit RTN rule + repack of the oracle (rounding
    rule parity-unpinned), sym and asym, nn.Linear layout."""
    rng = np.random.default_rng(24)
    w = (rng.standard_normal((96, 256)) * 0.05).astype(np.float32)  # [N, K]
    for asym in (False, True):
        blob = qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, 64, "fp32", "int8", "fp32", asym)
        q, s, z = orc.rtn_quantize_int8(w, True, 64, asym)
        assert np.array_equal(blob.cpu().numpy().view(np.uint8), orc.repack_int8(q, s, z, None, 64))
        deq = torch.empty(96, 256, dtype=torch.float32, device="cuda")
        qbits.dequantize_packed_weight(blob, deq, True, "fp32", "int8", "fp32")
        assert np.abs(deq.cpu().numpy() - w).max() <= 0.51 * s.max()  # within half a quantisation step


# ---- int3_clip / int2_clip (reference weight-type strings): narrow integers in int4 storage ------------------------
@pytest.mark.parametrize("wname,group,asym,shuf", [("int3_clip", 128, False, False), ("int3_clip", 32, True, True),
                                                   ("int2_clip", 64, True, False), ("int2_clip", -1, False, False)])
def test_narrow_int_blob_dequant_info_and_linear(qbits, wname, group, asym, shuf):
    """User-supplied int3 / int2 values pass through the blob exactly (the reference's packq contract,
    qbits_ut/test_packq.py:100-109): blob bytes = the oracle's int4 blob + the narrow_bits tag, same size as int4,
    dequantize = (q - zp) * scale, acquire_packed_weight_info keeps the name; woq_linear (decode GEMV, small batch,
    MFMA GEMM) against the oracle at the int4 bounds."""
    bits = orc.NARROW_BITS[wname]
    K, N, lim = 384, 80, 1 << (bits - 1)
    rng = np.random.default_rng(40 + bits)
    g = K if group == -1 else group
    G = K // g
    q = rng.integers(-lim, lim, (K, N), dtype=np.int8)
    s = (rng.random((G, N), dtype=np.float32) * 0.02 + 0.005).astype(np.float32)
    z = rng.integers(-lim, lim, (G, N), dtype=np.int8) if asym else None
    idx = rng.permutation(np.repeat(np.arange(G), g)).astype(np.int32) if shuf else None
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    blob = qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                         e8 if z is None else torch.from_numpy(z).cuda(),
                                         e32 if idx is None else torch.from_numpy(idx).cuda(), wname, "fp32", "fp32",
                                         asym, group)
    ref_blob = orc.repack_narrow(q, s, z, _cvt(idx, K, group), group, bits=bits)
    got = blob.cpu().numpy().view(np.uint8)
    assert got.size == qbits.get_packed_weight_size(K, N, wname, "fp32", "fp32", asym, group, shuf) \
        == qbits.get_packed_weight_size(K, N, "int4_clip", "fp32", "fp32", asym, group, shuf)
    assert np.array_equal(got, ref_blob)
    deq = torch.empty(K, N, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, deq, False, "fp32", wname, "fp32")
    assert np.array_equal(deq.cpu().numpy(), orc.dequant_raw(q, s, z, g))
    assert "".join(chr(c) for c in qbits.acquire_packed_weight_info(blob, 6).tolist()) == wname
    w = np.abs(orc.dequant_raw(q, s, z, g))
    for M, compute in ((1, "fp32"), (5, "fp32"), (64, "fp32"), (64, "bf16")):
        x = rng.standard_normal((M, K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        ref = orc.woq_linear(x, ref_blob, bias)
        out = torch.zeros(M, N, device="cuda")
        blob_c = blob if compute == "fp32" else qbits.repack_quantized_weight(
            torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(), e8 if z is None else torch.from_numpy(z).cuda(),
            e32 if idx is None else torch.from_numpy(idx).cuda(), wname, "fp32", compute, asym, group)
        qbits.woq_linear(torch.from_numpy(x).cuda(), blob_c, torch.from_numpy(bias).cuda(), out, compute, wname, "fp32",
                         asym)
        xs = np.abs(x if idx is None else x[:, _cvt(idx, K, group)])
        mag = xs @ w + np.abs(bias)
        rel = 2e-5 if (M <= 8 or compute == "fp32") else 2e-3
        assert (np.abs(out.cpu().numpy() - ref) <= rel * mag + 1e-5).all(), (M, compute)


@pytest.mark.parametrize("wname", ["int3_clip", "int2_clip"])
def test_narrow_int_quantize_to_packed_weight_roundtrip(qbits, wname):
    """qbits.quantize_to_packed_weight on the narrow widths == the oracle's RTN rule at that width + repack (rounding
    rule parity-unpinned), sym and asym, nn.Linear layout; dequantised within half a step (one step at
    the clipped top code of the symmetric range)."""
    bits = orc.NARROW_BITS[wname]
    rng = np.random.default_rng(50 + bits)
    w = (rng.standard_normal((96, 256)) * 0.05).astype(np.float32)  # [N, K]
    for asym in (False, True):
        blob = qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, 64, "fp32", wname, "fp32", asym)
        q, s, z = orc.rtn_quantize_bits(w, True, 64, asym, bits)
        assert np.array_equal(blob.cpu().numpy().view(np.uint8), orc.repack_narrow(q, s, z, None, 64, bits=bits))
        deq = torch.empty(96, 256, dtype=torch.float32, device="cuda")
        qbits.dequantize_packed_weight(blob, deq, True, "fp32", wname, "fp32")
        assert np.abs(deq.cpu().numpy() - w).max() <= 0.51 * s.max()


# ---- edge cases: empty / boundary batch sizes, dispatch seams, long K ----------------------------------------------
def test_woq_linear_empty_batch_is_a_noop(qbits):
    """M = 0 (an empty activation batch) returns without touching the output, like the reference's GEMM with m = 0."""
    q, s, z, idx = _mk(256, 48, 32, True, False, seed=31)
    blob = _gpu_blob(qbits, q, s, z, idx, 32)
    out = torch.empty(0, 48, device="cuda")
    qbits.woq_linear(torch.empty(0, 256, device="cuda"), blob, torch.empty(0), out, "fp32", "int4_clip", "fp32", True)
    assert out.shape == (0, 48)


@pytest.mark.parametrize("M", [4, 5, 8, 9, 16, 17, 127, 128, 129])
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_woq_linear_dispatch_seams(qbits, M, compute):
    """Row counts on both sides of every kernel switch: 4 | 5, 8 | 9 (MFMA row sets of the decode GEMV), 16 | 17 (GEMV -> MFMA GEMM),
    127 / 128 / 129 (GEMM row-block edge). Same bound as the route's own test."""
    K, N, group = 384, 80, 128
    q, s, z, idx = _mk(K, N, group, True, False, seed=32)
    blob = qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                         torch.from_numpy(z).cuda(), torch.empty(0, dtype=torch.int32), "int4_clip",
                                         "fp32", compute, True, group)
    rng = np.random.default_rng(33)
    x = rng.standard_normal((M, K)).astype(np.float32)
    ref = orc.woq_linear(x, orc.repack(q, s, z, None, group), None)
    out = torch.zeros(M, N, device="cuda")
    qbits.woq_linear(torch.from_numpy(x).cuda(), blob, torch.empty(0), out, compute, "int4_clip", "fp32", True)
    rel = 1e-4 if (compute == "fp32" or M <= 16) else 2e-3
    assert (np.abs(out.cpu().numpy() - ref) <= rel * np.abs(ref).max(axis=1, keepdims=True) + 1e-5).all()


@pytest.mark.parametrize("K", [16384, 16512])
def test_decode_gemv_longest_k(qbits, K):
    """K = 16384 is the longest contraction the tile kernel takes (16 waves x 8 tiles); one tile more goes to the
    generic kernel. Both against the oracle."""
    N, group = 32, 128
    q, s, z, idx = _mk(K, N, group, False, False, seed=34)
    blob = _gpu_blob(qbits, q, s, None, None, group)
    x = np.random.default_rng(35).standard_normal((1, K)).astype(np.float32)
    ref = orc.woq_linear(x, orc.repack(q, s, None, None, group), None)
    out = torch.zeros(1, N, device="cuda")
    qbits.woq_linear(torch.from_numpy(x).cuda(), blob, torch.empty(0), out, "fp32", "int4_clip", "fp32", False)
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-5


def test_shape_and_type_errors_keep_the_reference_prefix(qbits):
    """Every rejection surfaces as RuntimeError with the reference's 'QBits:' / 'Qbits:' prefix
    (bestla_weightonly_dispatcher.cpp:289,368; qbits.cpp:35,150)."""
    q, s, z, idx = _mk(256, 48, 32, False, False, seed=36)
    blob = _gpu_blob(qbits, q, s, None, None, 32)
    with pytest.raises(RuntimeError, match="QBits: woq_linear shape mismatch"):
        qbits.woq_linear(torch.zeros(2, 255, device="cuda"), blob, torch.empty(0), torch.zeros(2, 48, device="cuda"),
                         "fp32", "int4_clip", "fp32", False)
    with pytest.raises(RuntimeError, match="unsupported qbits data type"):
        qbits.woq_linear(torch.zeros(2, 256, device="cuda", dtype=torch.float64), blob, torch.empty(0),
                         torch.zeros(2, 48, device="cuda"), "fp32", "int4_clip", "fp32", False)
    with pytest.raises(RuntimeError, match="[Qq]bits: unsupported bestla packq config"):
        qbits.get_packed_weight_size(256, 48, "int5_clip", "fp32", "fp32", False, 32, False)
    with pytest.raises(RuntimeError, match="QBits: unsupported blocksize"):
        qbits.get_packed_weight_size(256, 48, "int4_clip", "fp32", "fp32", False, 48, False)
    with pytest.raises(RuntimeError, match="QBits: not a WQH1 packed weight"):
        qbits.woq_linear(torch.zeros(2, 256, device="cuda"), torch.zeros(4096, dtype=torch.int8, device="cuda"),
                         torch.empty(0), torch.zeros(2, 48, device="cuda"), "fp32", "int4_clip", "fp32", False)


# ---- 4-bit table weight types: nf4, fp4_e2m1, fp4_e2m1_bnb (reference strings, bestla_weightonly_dispatcher.hpp:62-70) --
TABLE_TYPES = {"nf4": orc.W_NF4, "fp4_e2m1": orc.W_FP4_E2M1, "fp4_e2m1_bnb": orc.W_FP4_E2M1_BNB}


@pytest.mark.parametrize("wname", sorted(TABLE_TYPES))
@pytest.mark.parametrize("K,N,group", [(512, 1024, 128), (256, 48, 32), (160, 24, 64), (512, 64, -1), (384, 272, 128)])
def test_table_weight_types_quantize_dequant_linear(qbits, wname, K, N, group):
    """w = table[code] * scale. quantize_to_packed_weight == the oracle's nearest-entry RTN + repack byte for byte
    (rounding rule parity-unpinned); dequantisation bit-exact; woq_linear at decode row counts (the
    generic fp32 kernel) and at prefill row counts (round 3: the MFMA GEMM over a pre-dequantised fragment image of the
    weight, hi + lo fp16 planes for compute fp32 — one and several row blocks, even and odd K-tile counts, ragged N)
    within fp32 summation error of dequantise -> matmul -> + bias; asym is rejected like the reference does for float
    weight types."""
    wt = TABLE_TYPES[wname]
    rng = np.random.default_rng(41)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)  # nn.Linear layout
    blob = qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, group, "fp32", wname, "fp32", False)
    q, s = orc.rtn_quantize_table(w, True, group, wt)
    ref_blob = orc.repack_table(q, s, wt, group)
    assert np.array_equal(blob.cpu().numpy().view(np.uint8), ref_blob)
    deq = torch.empty(K, N, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, deq, False, "fp32", wname, "fp32")
    want = orc.dequantize_blob(ref_blob)
    assert np.array_equal(deq.cpu().numpy(), want)
    assert "".join(chr(c) for c in qbits.acquire_packed_weight_info(blob, 6).tolist()) == wname
    bias = rng.random(N, dtype=np.float32)
    for M in (1, 3, 37, 300):
        x = rng.standard_normal((M, K)).astype(np.float32)
        ref = orc.woq_linear(x, ref_blob, bias)
        out = torch.zeros(M, N, device="cuda")
        qbits.woq_linear(torch.from_numpy(x).cuda(), blob, torch.from_numpy(bias).cuda(), out, "fp32", wname, "fp32",
                         False)
        mag = np.abs(x) @ np.abs(want)
        assert (np.abs(out.cpu().numpy() - ref) <= 2e-6 * mag + 1e-5).all()
    # reduced-precision compute (one fp16 product per operand pair), 16-bit rows in, 16-bit out
    blob16 = qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, group, "bf16", wname, "fp32", False)
    x = torch.from_numpy(rng.standard_normal((150, K)).astype(np.float32)).cuda().half()
    out16 = torch.zeros(150, N, device="cuda", dtype=torch.float16)
    qbits.woq_linear(x, blob16, torch.from_numpy(bias).cuda(), out16, "bf16", wname, "fp32", False)
    ref = orc.woq_linear(x.float().cpu().numpy(), ref_blob, bias)
    assert (np.abs(out16.float().cpu().numpy() - ref) <= 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 1e-3).all()
    with pytest.raises(RuntimeError, match="symmetric"):
        qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, group, "fp32", wname, "fp32", True)


# ---- fp8 weight types: fp8_e4m3, fp8_e5m2 (+ fp8_e8m0 scales) — reference strings, qbits_ut/test_weightonly.py:20-27 -----
FP8_TYPES = {"fp8_e4m3": orc.W_FP8_E4M3, "fp8_e5m2": orc.W_FP8_E5M2}


@pytest.mark.parametrize("wname,sname", [("fp8_e4m3", "fp32"), ("fp8_e4m3", "fp8_e8m0"), ("fp8_e5m2", "fp32"),
                                         ("fp8_e5m2", "fp8_e8m0")])
@pytest.mark.parametrize("K,N,group", [(512, 256, 128), (256, 48, 32), (160, 24, 64), (384, 64, -1)])
def test_fp8_weight_types_quantize_dequant_linear(qbits, wname, sname, K, N, group):
    """w = value(code) * scale on the OCP fp8 grids. quantize_to_packed_weight == the oracle's nearest-finite-code RTN
    (power-of-two scales for fp8_e8m0) + the two-plane composite blob, byte for byte (rounding rule parity-unpinned,
    DESIGN.md §4); dequantisation bit-exact; both names survive acquire_packed_weight_info; woq_linear at decode and
    at prefill row counts, fp32 and bf16 activations, within fp32 summation error of dequantise -> matmul -> + bias;
    asym and an fp8_e8m0 scale on a non-fp8 weight are rejected (the reference's validity matrix)."""
    wt, e8 = FP8_TYPES[wname], sname == "fp8_e8m0"
    rng = np.random.default_rng(61)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)  # nn.Linear layout
    blob = qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, group, "fp32", wname, sname, False)
    q, s = orc.rtn_quantize_fp8(w, True, group, wt, e8)
    ref_blob = orc.repack_fp8(q, s, wt, None, group, e8m0=e8)
    got = blob.cpu().numpy().view(np.uint8)
    assert got.size == ref_blob.size == qbits.get_packed_weight_size(K, N, wname, sname, "fp32", False, group, False)
    assert np.array_equal(got, ref_blob)
    deq = torch.empty(K, N, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, deq, False, "fp32", wname, sname)
    want = orc.dequantize_blob(ref_blob)
    assert np.array_equal(deq.cpu().numpy(), want)
    deq_t = torch.empty(N, K, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, deq_t, True, "fp32", wname, sname)
    assert np.array_equal(deq_t.cpu().numpy(), want.T)
    name = lambda t: "".join(chr(c) for c in qbits.acquire_packed_weight_info(blob, t).tolist())  # noqa: E731
    assert name(6) == wname and name(8) == sname
    g = K if group == -1 else group
    assert np.array_equal(qbits.acquire_packed_weight_info(blob, 9).cpu().numpy(), s)
    # relative error of the grid: half a ulp of a 3- / 2-bit mantissa on the group's scale (twice that with e8m0)
    assert np.abs(want - w.T).max() <= (2.0 ** -4 if wt == orc.W_FP8_E4M3 else 2.0 ** -3) * np.abs(w).max() * (2 if e8 else 1)
    bias = rng.random(N, dtype=np.float32)
    for M, adt in ((1, torch.float32), (3, torch.float32), (37, torch.float32), (5, torch.bfloat16),
                   (260, torch.float32)):  # 37 / 260 rows: the MFMA GEMM over the pre-dequantised fragment image
        x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(adt)
        xf = x.float().numpy()
        ref = orc.woq_linear(xf, ref_blob, bias)
        out = torch.zeros(M, N, device="cuda")
        qbits.woq_linear(x.cuda(), blob, torch.from_numpy(bias).cuda(), out, "fp32", wname, sname, False)
        mag = np.abs(xf) @ np.abs(want)
        assert (np.abs(out.cpu().numpy() - ref) <= 2e-6 * mag + 1e-5).all(), M
    with pytest.raises(RuntimeError, match="symmetric"):
        qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, group, "fp32", wname, sname, True)
    with pytest.raises(RuntimeError, match="fp8_e8m0"):
        qbits.quantize_to_packed_weight(torch.from_numpy(w).cuda(), True, group, "fp32", "int4_clip", "fp8_e8m0", False)


@pytest.mark.parametrize("wname", sorted(FP8_TYPES))
def test_fp8_repack_passes_codes_through_with_act_shuffle(qbits, wname):
    """User-supplied code bytes (every finite code), scales and a GPTQ g_idx pass through the composite blob exactly
    (the packq contract, qbits_ut/test_packq.py:100-109): blob bytes = oracle, g_idx comes back as the shuffle
    indices, woq_linear applies the activation shuffle."""
    wt = FP8_TYPES[wname]
    K, N, group = 256, 80, 64
    rng = np.random.default_rng(62)
    finite = np.flatnonzero(np.isfinite(orc.FP8_TABLES[wt])).astype(np.uint8)
    codes = finite[rng.integers(0, finite.size, (K, N))]
    codes[:finite.size // N * N].flat[:finite.size] = finite  # every finite code at least once
    s = (rng.random((K // group, N), dtype=np.float32) * 1e-3 + 1e-4).astype(np.float32)
    idx = rng.permutation(np.arange(K, dtype=np.int32) // group).astype(np.int32)
    blob = qbits.repack_quantized_weight(torch.from_numpy(codes.view(np.int8)).cuda(), torch.from_numpy(s).cuda(),
                                         torch.empty(0, dtype=torch.int8), torch.from_numpy(idx).cuda(), wname, "fp32",
                                         "fp32", False, group)
    ref_blob = orc.repack_fp8(codes, s, wt, _cvt(idx, K, group), group)
    assert np.array_equal(blob.cpu().numpy().view(np.uint8), ref_blob)
    assert np.array_equal(orc.fp8_codes_of(ref_blob), codes)
    assert np.array_equal(qbits.acquire_packed_weight_info(blob, 5).cpu().numpy(), _cvt(idx, K, group))
    deq = torch.empty(K, N, dtype=torch.float32, device="cuda")
    qbits.dequantize_packed_weight(blob, deq, False, "fp32", wname, "fp32")
    want = orc.FP8_TABLES[wt][codes] * np.repeat(s, group, axis=0)
    assert np.array_equal(deq.cpu().numpy(), want)
    x = rng.standard_normal((6, K)).astype(np.float32)
    ref = orc.woq_linear(x, ref_blob, None)
    out = torch.zeros(6, N, device="cuda")
    qbits.woq_linear(torch.from_numpy(x).cuda(), blob, torch.empty(0), out, "fp32", wname, "fp32", False)
    mag = np.abs(x)[:, _cvt(idx, K, group)] @ np.abs(want)
    assert (np.abs(out.cpu().numpy() - ref) <= 2e-6 * mag + 1e-5).all()
