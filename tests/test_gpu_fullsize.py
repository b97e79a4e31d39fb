"""Full-size (BASELINE.json configs) properties of the HIP path that need no oracle run at that size:
linearity, dequantize <-> linear consistency on one-hot probes, decode-vs-prefill agreement, blob round trip,
engine determinism. The CPU oracle would take minutes per matrix at these sizes; these properties do not."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LLAMA7B = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]  # fused qkv, o, fused gate/up, down


@pytest.fixture(scope="module")
def qbits():
    from intel_extension_for_transformers_amd import qbits as q

    return q


def _rand_blob(qbits, K, N, group, asym, seed, scale_dtype="fp16", compute="fp32"):
    g = torch.Generator(device="cuda").manual_seed(seed)
    G = 1 if group == -1 else K // group
    q = torch.randint(-8, 8, (K, N), generator=g, device="cuda", dtype=torch.int8)
    s = (torch.rand(G, N, generator=g, device="cuda") + 0.5) * 0.005
    z = torch.randint(-8, 8, (G, N), generator=g, device="cuda", dtype=torch.int8) if asym else torch.empty(0, dtype=torch.int8)
    blob = qbits.repack_quantized_weight(q, s, z, torch.empty(0, dtype=torch.int32), "int4_clip", scale_dtype, compute,
                                         asym, group)
    return blob, q, s, z


@pytest.mark.parametrize("K,N", LLAMA7B)
@pytest.mark.parametrize("group,asym", [(128, False), (32, True)])
def test_fullsize_decode_matches_dequantised_matmul(qbits, K, N, group, asym):
    """The reference's own criterion (qbits_ut/test_weightonly.py:51-88) at Llama-2-7B size: woq_linear(M=1) against
    dequantize_packed_weight -> fp64 matmul. Bound 1e-4 * max|ref| (the reference allows rtol 0.03)."""
    blob, _, _, _ = _rand_blob(qbits, K, N, group, asym, seed=K + N)
    x = torch.randn(1, K, device="cuda")
    out = torch.empty(1, N, device="cuda")
    qbits.woq_linear(x, blob, torch.empty(0), out, "fp32", "int4_clip", "fp16", asym)
    w = torch.empty(K, N, device="cuda")
    qbits.dequantize_packed_weight(blob, w, False, "fp32", "int4_clip", "fp16")
    ref = (x.double() @ w.double()).float()
    assert (out - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("K,N", LLAMA7B)
def test_fullsize_linearity_and_one_hot(qbits, K, N):
    """f(a x + b y) = a f(x) + b f(y) within fp32 rounding, and f(e_k) = row k of the dequantised weight exactly
    (a one-hot activation is represented exactly by the fixed-point limbs, and every partial sum is an integer)."""
    blob, _, _, _ = _rand_blob(qbits, K, N, 128, False, seed=3)
    e = torch.empty(0)
    f = lambda v: (lambda o: (qbits.woq_linear(v, blob, e, o, "fp32", "int4_clip", "fp16", False), o)[1])(  # noqa: E731
        torch.empty(v.shape[0], N, device="cuda"))
    x, y = torch.randn(1, K, device="cuda"), torch.randn(1, K, device="cuda")
    lhs = f(2.0 * x - 0.5 * y)
    rhs = 2.0 * f(x) - 0.5 * f(y)
    assert (lhs - rhs).abs().max().item() <= 2e-5 * rhs.abs().max().item() + 1e-6
    w = torch.empty(K, N, device="cuda")
    qbits.dequantize_packed_weight(blob, w, False, "fp32", "int4_clip", "fp16")
    for k in (0, 1, 127, 128, K // 2 + 5, K - 1):
        oh = torch.zeros(1, K, device="cuda")
        oh[0, k] = 1.0
        assert torch.equal(f(oh)[0], w[k])


@pytest.mark.parametrize("K,N", [(4096, 12288), (11008, 4096)])
def test_fullsize_prefill_agrees_with_decode(qbits, K, N):
    """Row m of the MFMA GEMM (M = 160) == the small-M kernel on that row: two different kernels, same contract."""
    blob, _, _, _ = _rand_blob(qbits, K, N, 128, True, seed=5)
    x = torch.randn(160, K, device="cuda")
    e = torch.empty(0)
    big = torch.empty(160, N, device="cuda")
    qbits.woq_linear(x, blob, e, big, "fp32", "int4_clip", "fp16", True)
    for m in (0, 77, 159):
        one = torch.empty(1, N, device="cuda")
        qbits.woq_linear(x[m:m + 1].contiguous(), blob, e, one, "fp32", "int4_clip", "fp16", True)
        assert (big[m] - one[0]).abs().max().item() <= 1e-4 * one.abs().max().item()


def test_fullsize_repack_roundtrip(qbits):
    """repack -> acquire_packed_weight_info / dequantize returns the integers, scales and zero points exactly."""
    K, N, group = 11008, 4096, 128
    blob, q, s, z = _rand_blob(qbits, K, N, group, True, seed=9, scale_dtype="fp32")
    assert torch.equal(qbits.acquire_packed_weight_info(blob, 9), s)
    assert torch.equal(qbits.acquire_packed_weight_info(blob, 10), z)
    w = torch.empty(K, N, device="cuda")
    qbits.dequantize_packed_weight(blob, w, False, "fp32", "int4_clip", "fp32")
    rows = torch.arange(K, device="cuda") // group
    ref = (q.float() - z[rows].float()) * s[rows]
    assert torch.equal(w, ref)


def test_fullsize_engine_is_deterministic_and_graph_equals_eager():
    """Llama-2-7B-shaped engine (4 layers to keep the test short): two identical runs give identical logits bit for
    bit (integer tile sums, fixed reduction order), and hipGraph replay reproduces the eager token chain."""
    from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights

    def run(graph):
        eng = WoqDecoderEngine(4096, 11008, 32, 32, 128, 4, 32000, max_ctx=64)
        synth_llama_weights(eng, 4096, 11008, 32, 32, 128, 4, 32000, group=128, sym=True, scale_dtype="fp16", seed=11)
        for i, t in enumerate([11, 222, 3333]):
            eng.token.fill_(t)
            eng.pos.fill_(i)
            eng.step(greedy=(i == 2))
        first = eng.logits.clone()
        toks = [int(eng.token.item())]
        if graph:
            eng.capture(greedy=True)
        for _ in range(5):
            eng.replay(1) if graph else eng.step(greedy=True)
            toks.append(int(eng.token.item()))
        return first, toks

    l0, t0 = run(False)
    l1, t1 = run(True)
    assert torch.equal(l0, l1) and t0 == t1
    assert torch.isfinite(l0).all()


def test_llama70b_tp8_rank_shard_shapes():
    """BASELINE configs[3]: the per-rank shapes of Llama-2-70B at TP = 8 (hidden 8192, 8 q heads / 1 kv head per
    rank, inter 3584 per rank, vocab shard 4000) on ONE device, 2 layers, synthetic weights: the decode GEMV
    geometry (K = 8192 -> 64 k-tiles, qkv N = 1280, row-parallel o / down with K = 1024 / 3584) and the prompt-pass
    GEMMs must agree with each other — prefill of a prompt vs token-by-token decode of the same prompt, last-position
    logits within the prompt-pass tolerance (1e-2 relative), then identical greedy continuation lengths. Rank 0 of a
    one-rank group: the partial sums are the full sums, so this checks kernels and shapes, not collectives."""
    from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights

    dims = dict(hidden=8192, inter=3584, heads=8, kv_heads=1, head_dim=128, layers=2, vocab=4000)

    def make():
        e = WoqDecoderEngine(dims["hidden"], dims["inter"], dims["heads"], dims["kv_heads"], dims["head_dim"],
                             dims["layers"], dims["vocab"], max_ctx=256)
        synth_llama_weights(e, dims["hidden"], dims["inter"], dims["heads"], dims["kv_heads"], dims["head_dim"],
                            dims["layers"], dims["vocab"], group=128, sym=True, scale_dtype="fp16", seed=77)
        return e

    g = torch.Generator().manual_seed(3)
    prompt = torch.randint(0, dims["vocab"], (140,), generator=g).tolist()
    a, b = make(), make()
    la = a.prefill(prompt, greedy=False)[0].float().cpu()
    for i, t in enumerate(prompt):
        b.token.fill_(t)
        b.pos.fill_(i)
        b.step(greedy=False)
    lb = b.logits.float().cpu()
    assert torch.isfinite(la).all() and torch.isfinite(lb).all()
    assert (la - lb).abs().max().item() <= 1e-2 * lb.abs().max().item() + 1e-3
