"""Pin the CPU oracle against outputs of the reference's own code (tests/golden/make_golden.py).

Integer / byte / index work is compared bit-exact; floating point with the tolerance written
at each assert.
"""
import os

import numpy as np
import pytest

from oracle import woq_oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("tag", ["b4_sym", "b4_asym", "b4_asym_g128", "b8_sym", "b8_asym"])
def test_unpack_weight_matches_reference(golden_dir, tag):
    """reference: llm/quantization/utils.py:82-125 executed verbatim by make_golden.py."""
    g = _load(golden_dir, "unpack_weight.npz")
    K, N, group, bits, sym = [int(v) for v in g[f"{tag}_meta"]]
    w, z = orc.unpack_weight(g[f"{tag}_qweight"], g[f"{tag}_qzeros"], K, N, K // group, bits, bool(sym))
    assert np.array_equal(w.astype(np.int16), g[f"{tag}_w"])  # bit-exact
    assert np.array_equal(z.astype(np.int16), g[f"{tag}_z"])  # bit-exact (includes the zp-1 "+1" quirk)


@pytest.mark.parametrize("tag", ["rtn_sym", "rtn_asym"])
def test_signed_nibble_convention(golden_dir, tag):
    """reference: nn/modules.py:225-237 — what set_weights_bias hands to repack_quantized_weight."""
    g = _load(golden_dir, "set_weights_bias.npz")
    w = orc.to_signed_nibble(g[f"{tag}_in_w"].astype(np.int8))
    assert np.array_equal(w.astype(np.int16), g[f"{tag}_out_w"])
    assert w.min() >= -8 and w.max() <= 7
    if int(g[f"{tag}_out_asym"]):
        z = orc.to_signed_nibble(g[f"{tag}_in_z"].astype(np.int8))
        assert np.array_equal(z.astype(np.int16), g[f"{tag}_out_z"])
    else:
        assert g[f"{tag}_out_z"].size == 0  # sym drops the zero points (modules.py:233-234)
    assert np.array_equal(g[f"{tag}_in_s"], g[f"{tag}_out_s"])  # scales only up-cast to fp32
    assert g[f"{tag}_out_gidx"].size == 0  # non-GPTQ drops g_idx (modules.py:236-237)


def test_convert_idx_matches_reference(golden_dir):
    """reference: qbits/qbits_ut/test_packq.py:22-28."""
    g = _load(golden_dir, "convert_idx.npz")
    assert np.array_equal(orc.convert_idx(g["ut_gidx"], 512, 128), g["ut_ret"])
    assert np.array_equal(orc.convert_idx(g["rnd_gidx"], 256, 32), g["rnd_ret"])


def test_gptq_desc_act_row_permutation(golden_dir):
    """modules.py:205-224: with desc_act the int weight rows are re-ordered group-sorted; that
    order is exactly convert_idx's shuffle order (row j of the packed weight = original row idx[j])."""
    g = _load(golden_dir, "set_weights_bias.npz")
    gidx = g["gptq_desc_act_in_gidx"]
    idx = orc.convert_idx(gidx, gidx.size, 32)
    w_in = g["gptq_desc_act_in_w"].astype(np.int8)
    expect = orc.to_signed_nibble(w_in[idx])
    assert np.array_equal(expect.astype(np.int16), g["gptq_desc_act_out_w"])
    assert np.array_equal(g["gptq_desc_act_out_gidx"], gidx)  # the raw g_idx is forwarded to repack


def test_dequant_is_inverse_of_reference_requant(golden_dir):
    """reference: nn/modules.py:264-295 quant_weight_w_scale: q = round(w/scale + zp). Feeding the
    oracle's dequantised weights through the reference's own re-quantiser must reproduce q exactly
    (this is the round trip recover_qparms relies on, modules.py:356-372), tail group included."""
    g = _load(golden_dir, "quant_weight_w_scale.npz")
    group = int(g["group"])
    q_ref = g["q_asym"]  # [N, K] float, produced by the reference from (w, scale, zp)
    scale, zp = g["scale"], g["zp"].astype(np.int16)
    N, K = q_ref.shape
    # oracle works in [K, N] with [G, N] params and signed-domain values
    q_clamped = np.clip(q_ref, 0, 15)
    qs = (q_clamped.T - 8).astype(np.int8)
    zps = (zp.T - 8).astype(np.int8)
    W = orc.dequant_raw(qs, scale.T.copy(), zps, group)  # (q - zp) * s
    # reference formula on the oracle's output: round(W/scale + zp) == q
    G = scale.shape[1]
    kk = np.minimum(np.arange(K) // group, G - 1)
    req = np.rint(W / scale.T[kk, :] + zp.T[kk, :]).T
    assert np.array_equal(req, q_clamped)


def test_hf_ops(golden_dir):
    """RMSNorm / RoPE / SiLU*mul / GeLU against HF transformers (what the reference executes
    between its quantised linears). fp32; tolerance 2e-6 relative + 2e-6 absolute (libm vs torch)."""
    g = _load(golden_dir, "hf_ops.npz")
    y = orc.rmsnorm(g["rms_x"], g["rms_w"], float(g["rms_eps"]))
    np.testing.assert_allclose(y, g["rms_y"], rtol=2e-6, atol=2e-6)
    q, k, pos = g["rope_q"][0], g["rope_k"][0], g["rope_pos"][0]  # [heads, tokens, D]
    qr = orc.rope(q.transpose(1, 0, 2), pos)
    kr = orc.rope(k.transpose(1, 0, 2), pos)
    # large positions (4095) amplify fp32 angle rounding: 2e-4 absolute on O(1) values
    np.testing.assert_allclose(qr.transpose(1, 0, 2), g["rope_qr"][0], rtol=0, atol=2e-4)
    np.testing.assert_allclose(kr.transpose(1, 0, 2), g["rope_kr"][0], rtol=0, atol=2e-4)
    np.testing.assert_allclose(orc.silu_mul(g["silu_gate"], g["silu_up"]), g["silu_y"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(orc.gelu(g["gelu_x"], "tanh"), g["gelu_new_y"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(orc.gelu(g["gelu_x"], "erf"), g["gelu_y"], rtol=2e-6, atol=2e-6)


# ---- self-consistency of the blob restatement (the reference's own test idiom, F7) -------------
@pytest.mark.parametrize("K,N,group,asym,shuf", [
    (512, 1024, 128, False, False),   # qbits_ut/test_weightonly.py shape
    (512, 1024, 128, True, True),     # qbits_ut/test_packq.py shape
    (512, 1024, -1, False, False),    # blocksize -1 -> K (dispatcher.cpp:296)
    (256, 48, 32, True, False),       # scale_mode 1 (expanded per 32-block)
    (160, 24, 64, True, False),       # tail group + K padding + group 64
    (96, 20, 32, False, False),       # N padding
    (16, 8, -1, False, False),        # tiny, single group
])
def test_blob_roundtrip(K, N, group, asym, shuf):
    rng = np.random.default_rng(0)
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    g = K if group == -1 else group
    G = (K + g - 1) // g
    scales = rng.random((G, N), dtype=np.float32)
    zp = rng.integers(-4, 4, (G, N), dtype=np.int8) if asym else None
    idx = rng.permutation(K).astype(np.int32) if shuf else None
    blob = orc.repack(q, scales, zp, idx, group)
    h = orc.header(blob)
    assert h["total_bytes"] == blob.size == orc.packed_size(K, N, group, orc.F32, asym, shuf)
    assert (h["K"], h["N"], h["group"]) == (K, N, g)
    W = orc.dequantize_blob(blob)
    assert np.array_equal(W, orc.dequant_raw(q, scales, zp, g))  # exact: same fp32 product
    assert np.array_equal(orc.dequantize_blob(blob, transpose=True), W.T)
    x = rng.random((7, K), dtype=np.float32)
    ref = (x[:, idx] if shuf else x).astype(np.float64) @ W.astype(np.float64)
    out = orc.woq_linear(x, blob)
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)
    bias = rng.random(N, dtype=np.float32) * 10
    np.testing.assert_allclose(orc.woq_linear(x, blob, bias), ref + bias, rtol=1e-6, atol=1e-5)
    # streaming fp32 port (cpu_baseline leg) agrees with the definition: fp32 accumulate over K
    gemv = orc.woq_gemv_stream(x[0], blob, bias)
    np.testing.assert_allclose(gemv, ref[0] + bias, rtol=2e-5, atol=2e-4)


def test_reference_unit_test_idiom():
    """qbits_ut/test_weightonly.py:51-88 restated on the oracle: seed 0, uniform [0,1) inputs,
    m=256 n=1024 k=512, quantize -> dequantize -> matmul, allclose(rtol=0.03)."""
    import torch

    torch.manual_seed(0)
    act = torch.rand(256, 512)
    raw = torch.rand(512, 1024)
    for group, asym in [(128, False), (128, True), (-1, False)]:
        q, s, z = orc.rtn_quantize(raw.numpy(), False, group, asym)
        blob = orc.repack(q, s, z, None, group)
        revert = torch.from_numpy(orc.dequantize_blob(blob))
        ref = torch.matmul(act, revert)
        tar = torch.from_numpy(orc.woq_linear(act.numpy(), blob))
        assert torch.allclose(tar, ref, rtol=0.03)
        # and the RTN error itself is bounded: half a step (sym); asym on all-positive data clamps
        # zp at 0 so the top of the range saturates — allow 1.5 steps there
        g = 512 if group == -1 else group
        step = np.repeat(s, g, axis=0)
        assert np.all(np.abs(revert.numpy() - raw.numpy()) <= (1.5 if asym else 0.5) * step + 1e-6)


def test_scale_storage_types():
    rng = np.random.default_rng(1)
    K, N = 256, 32
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    s = rng.random((2, N), dtype=np.float32)
    for st in (orc.BF16, orc.F16):
        blob = orc.repack(q, s, None, None, 128, scale_type=st)
        W = orc.dequantize_blob(blob)
        s_r = orc.bf16_round(s) if st == orc.BF16 else s.astype(np.float16).astype(np.float32)
        assert np.array_equal(W, orc.dequant_raw(q, s_r, None, 128))


def test_unsupported_blocksize():
    q = np.zeros((96, 16), np.int8)
    with pytest.raises(RuntimeError):
        orc.repack(q, np.ones((6, 16), np.float32), None, None, 16)


@pytest.mark.parametrize("bits", [2, 3, 4, 8])
@pytest.mark.parametrize("asym", [False, True])
def test_rtn_rule_any_width(bits, asym):
    """The RTN restatement at every integer width the boundary names (int2_clip / int3_clip / int4_clip / int8): codes
    stay in the signed `bits` range, zero points too, dequantisation lands within half a step (one step where the
    symmetric range clips its top code), and the 4-bit case equals the C oracle's rule bit for bit."""
    rng = np.random.default_rng(7 + bits)
    w = (rng.standard_normal((96, 64)) * 0.05).astype(np.float32)
    q, s, z = orc.rtn_quantize_bits(w, False, 32, asym, bits)
    lim = 1 << (bits - 1)
    assert q.min() >= -lim and q.max() < lim and s.shape == (3, 64)
    assert (z is None) == (not asym) and (z is None or (z.min() >= -lim and z.max() < lim))
    deq = orc.dequant_raw(q, s, z, 32)
    step = np.repeat(s, 32, axis=0)
    assert (np.abs(deq - w) <= 0.5001 * step + 1e-7).all()
    if bits == 4:
        q4, s4, z4 = orc.rtn_quantize(w, False, 32, asym)
        assert np.array_equal(q, q4) and np.array_equal(s, s4) and (z is None or np.array_equal(z, z4))


def test_narrow_int_blob_is_the_int4_blob_plus_a_tag():
    rng = np.random.default_rng(3)
    q = rng.integers(-4, 4, (256, 48), dtype=np.int8)
    s = (rng.random((2, 48), dtype=np.float32) + 0.5).astype(np.float32)
    z = rng.integers(-4, 4, (2, 48), dtype=np.int8)
    b4, b3 = orc.repack(q, s, z, None, 128), orc.repack_narrow(q, s, z, None, 128, bits=3)
    assert orc.header(b3)["narrow_bits"] == 3 and orc.header(b4)["narrow_bits"] == 0
    assert orc.header(b3)["weight_type"] == orc.W_INT4
    b3.view(np.uint32)[15] = 0
    assert np.array_equal(b3, b4)
    assert np.array_equal(orc.dequantize_blob(b4), orc.dequant_raw(q, s, z, 128))
    with pytest.raises(AssertionError):
        orc.repack_narrow(q, s, z, None, 128, bits=2)


@pytest.mark.parametrize("wt,tname", [(orc.W_FP8_E4M3, "float8_e4m3fn"), (orc.W_FP8_E5M2, "float8_e5m2")])
def test_fp8_tables_and_rtn_against_torch_float8(wt, tname):
    """Pins the fp8 restatement on an independent implementation: the 256-entry value tables equal torch's float8
    dtypes code for code (NaN / inf codes included), and the nearest-finite-code rule reproduces torch's
    round-to-nearest-even conversion of w / scale wherever no exact tie occurs (ties: lowest code here)."""
    import torch

    dt = getattr(torch, tname)
    ref = torch.arange(256, dtype=torch.uint8).view(dt).float().numpy()
    tab = orc.FP8_TABLES[wt]
    assert np.array_equal(np.isnan(ref), np.isnan(tab)) and np.array_equal(ref[~np.isnan(ref)], tab[~np.isnan(tab)])
    assert np.nanmax(np.where(np.isfinite(tab), tab, np.nan)) == orc.FP8_MAX[wt]
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((48, 192)) * 0.05).astype(np.float32)
    for e8 in (False, True):
        q, s = orc.rtn_quantize_fp8(w, True, 64, wt, e8)
        v = (w.T / np.repeat(s, 64, axis=0)).astype(np.float32)
        tq = torch.from_numpy(v).to(dt).view(torch.uint8).numpy()
        assert (tq != q).mean() <= 1e-3  # exact ties only
        assert np.abs(v).max() <= orc.FP8_MAX[wt] * (1 + 2.0 ** -22)  # amax / (amax / max), two fp32 roundings
        if e8:
            assert np.all(np.frexp(s)[0] == 0.5)  # powers of two
            assert np.all(s >= np.abs(w.T).reshape(3, 64, 48).max(1) / np.float32(orc.FP8_MAX[wt]))
        blob = orc.repack_fp8(q, s, wt, None, 64, e8m0=e8)
        h = orc.header(blob)
        assert h["weight_type"] == wt and bool(h["flags"] & orc.FLAG_SCALE_E8M0) == e8
        assert h["scale_type"] == (orc.BF16 if e8 else orc.F32)
        assert np.array_equal(orc.fp8_codes_of(blob), q)
        assert np.array_equal(orc.dequantize_blob(blob), tab[q] * np.repeat(s, 64, axis=0))


@pytest.mark.parametrize("group,asym", [(128, False), (32, True), (96, True)])
def test_cpu_port_matches_the_oracle_definition(group, asym):
    """oracle/woq_cpu_port.c (the port bench.py times as cpu_baseline) against the oracle's definition of the same
    linear (dequantise -> matmul -> + bias, double accumulate): a port that computes something else would make the
    reported baseline meaningless. fp32 accumulation in k order per group -> 1e-5 * max|ref| is ample."""
    rng = np.random.default_rng(group)
    K, N = 768, 1000  # N not a multiple of the port's 32-column blocks; group 96: 8 whole groups
    G = (K + group - 1) // group
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    s = ((rng.random((G, N), dtype=np.float32) + 0.5) * 0.01).astype(np.float32)
    z = rng.integers(-8, 8, (G, N), dtype=np.int8) if asym else None
    x = rng.standard_normal(K).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = orc.woq_linear(x[None], orc.repack(q, s, z, None, group), b)[0]
    got = orc.CpuPortLinear(q, s, z, group)(x, b)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    assert orc.cpu_port_isa() in ("avx512", "avx2", "scalar")


def _tiny_oracle(window=0, kv_heads=2, seed=11):
    cfg = dict(hidden=128, inter=256, heads=4, kv_heads=kv_heads, head_dim=32, layers=2, vocab=96, eps=1e-5,
               theta=10000.0, window=window)
    rng = np.random.default_rng(seed)
    H, I, NH, KV, D = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    layers = []
    for _ in range(cfg["layers"]):
        ly = {}
        for n, (k, nn) in dict(q=(H, NH * D), k=(H, KV * D), v=(H, KV * D), o=(NH * D, H), gate=(H, I), up=(H, I),
                               down=(I, H)).items():
            q, s, z = orc.rtn_quantize(rng.standard_normal((k, nn)).astype(np.float32) * 0.08, False, 32, True)
            ly[n] = orc.repack(q, s, z, None, 32)
        ly["ln1"] = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        ly["ln2"] = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        layers.append(ly)
    embed = rng.standard_normal((cfg["vocab"], H)).astype(np.float32)
    lm = (rng.standard_normal((cfg["vocab"], H)) * 0.1).astype(np.float32)
    norm = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    return orc.LlamaOracle(cfg, embed, layers, norm, lm), cfg


@pytest.mark.parametrize("window", [0, 7])
def test_many_row_prompt_oracle_equals_the_token_loop(window):
    """`LlamaOracle.forward_prompt` (the many-row restatement the full-geometry attention parity tests use) is the
    token-by-token oracle: same logits at every position, one shot and in chunks, and `forward_token` continues on the
    cache it leaves — causal mask, sliding window and grouped-query heads included."""
    oracle, cfg = _tiny_oracle(window=window)
    rng = np.random.default_rng(3)
    prompt = rng.integers(0, cfg["vocab"], 23).tolist()
    ref = np.stack([oracle.forward_token(t, i) for i, t in enumerate(prompt)])
    kref = [k.copy() for k in oracle.k]
    oracle.reset()
    got = oracle.forward_prompt(prompt, all_logits=True)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    for a, b in zip(oracle.k, kref):
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    oracle.reset()
    for s0 in (0, 9, 18):
        last = oracle.forward_prompt(prompt[s0:s0 + 9], start_pos=s0)
    assert np.abs(last - ref[-1]).max() <= 2e-5 * np.abs(ref).max()
    nxt = int(ref[-1].argmax())
    a = oracle.forward_token(nxt, len(prompt))
    oracle.reset()
    for i, t in enumerate(prompt + [nxt]):
        b = oracle.forward_token(t, i)
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()


def test_linear_rows_applies_the_act_order_shuffle():
    """The oracle's many-row linear (prompt passes) gathers the activation columns by the blob's stored shuffle exactly
    like its row-at-a-time `woq_linear` (reference definition: `index_select(x, 1, g_idx)`, autograd/functions.py:48-50;
    qbits_ut/test_packq.py:71) — round 5 found it ignoring the shuffle, which only act-order prompt passes could see."""
    rng = np.random.default_rng(0)
    K, N, g = 256, 48, 64
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    s = (rng.random((K // g, N), dtype=np.float32) + 0.5) * 0.01
    z = rng.integers(-8, 8, (K // g, N), dtype=np.int8)
    g_idx = rng.permutation(np.arange(K, dtype=np.int32) // g).astype(np.int32)
    shuffle = orc.convert_idx(g_idx, K, g)
    blob = orc.repack(q, s, z, shuffle, g)
    x = rng.standard_normal((5, K)).astype(np.float32)
    want = x[:, shuffle].astype(np.float64) @ orc.dequant_raw(q, s, z, g).astype(np.float64)
    assert np.abs(orc.woq_linear(x, blob) - want).max() <= 1e-5
    assert np.abs(orc.linear_rows(x, blob) - want).max() <= 1e-5
    assert np.abs(orc.linear_rows(x, orc.repack(q, s, z, None, g)) - x.astype(np.float64) @ orc.dequant_raw(q, s, z, g)).max() <= 1e-5
