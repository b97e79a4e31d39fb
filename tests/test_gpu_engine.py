"""Decode-engine parity: fused HIP decode path vs the fp32 CPU oracle decoder on the SAME (q, scale, zp).

Tolerance (north_star: "logits max-abs"): the engine computes in fp32 with an fp16 KV cache and fp16
embedding / lm_head storage; the oracle is fp32 throughout on the same stored values. Stated bound:
max|logit_gpu - logit_oracle| <= 2e-3 * max|logit_oracle| + 1e-4, and greedy tokens identical.
"""
import numpy as np
import pytest
import torch

from oracle import woq_oracle as orc

pytestmark = pytest.mark.gpu


def _tiny(group, asym, scale_dtype, seed=0, max_ctx=64, max_batch=1, head_dim=64, kv_dtype=torch.float16,
          attn_splits=0, window=0, hidden=256, attn_grouped=False, weight_dtype="int4_clip", compute_dtype="fp32",
          act_order=False):
    from intel_extension_for_transformers_amd import qbits
    from intel_extension_for_transformers_amd.runtime import WoqDecoderEngine, fuse_gate_up

    cfg = dict(hidden=hidden, inter=512, heads=hidden // head_dim, kv_heads=128 // head_dim, head_dim=head_dim, layers=2,
               vocab=384, eps=1e-5, theta=10000.0, window=window)
    rng = np.random.default_rng(seed)
    eng = WoqDecoderEngine(cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"], cfg["layers"],
                           cfg["vocab"], max_ctx=max_ctx, rms_eps=cfg["eps"], rope_theta=cfg["theta"],
                           max_batch=max_batch, kv_dtype=kv_dtype, attn_splits=attn_splits, sliding_window=window,
                           attn_grouped=attn_grouped)
    st = {"fp32": orc.F32, "fp16": orc.F16, "bf16": orc.BF16}[scale_dtype]
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)

    wt = {"nf4": orc.W_NF4, "fp4_e2m1": orc.W_FP4_E2M1, "fp4_e2m1_bnb": orc.W_FP4_E2M1_BNB}.get(weight_dtype)
    f8 = {"fp8_e4m3": orc.W_FP8_E4M3, "fp8_e5m2": orc.W_FP8_E5M2}.get(weight_dtype)

    def quant(k, n):
        w = rng.standard_normal((k, n)).astype(np.float32) * 0.05
        if f8 is not None:  # fp8 code bytes (as int8), symmetric
            q, s = orc.rtn_quantize_fp8(w, False, group, f8)
            return q.view(np.int8), s, None
        if wt is not None:  # 4-bit table type: codes 0..15, symmetric
            return orc.rtn_quantize_table(w, False, group, wt) + (None,)
        return orc.rtn_quantize(w, False, group, asym)

    def gpu_pack(q, s, z, g_idx=None):
        return qbits.repack_quantized_weight(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(),
                                             e8 if z is None else torch.from_numpy(z).cuda(),
                                             e32 if g_idx is None else torch.from_numpy(g_idx).cuda(), weight_dtype,
                                             scale_dtype, compute_dtype, z is not None, group)

    def cpu_pack(q, s, z, g_idx=None):
        if f8 is not None:
            return orc.repack_fp8(q.view(np.uint8), s, f8, None, group, scale_type=st)
        if wt is not None:
            return orc.repack_table(q, s, wt, group, scale_type=st)
        shuf = None if g_idx is None else orc.convert_idx(g_idx, q.shape[0], q.shape[0] if group == -1 else group)
        return orc.repack(q, s, z, shuf, group, scale_type=st)

    def order_of(k):  # raw GPTQ g_idx of one (fused) projection: group id of every K row, act-order positions
        return rng.permutation(np.arange(k, dtype=np.int32) // (k if group == -1 else group)).astype(np.int32)

    H, I, NH, KV, D = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    layers = []
    for l in range(cfg["layers"]):
        ly = {}
        parts = {n: quant(k, nn) for n, (k, nn) in dict(q=(H, NH * D), k=(H, KV * D), v=(H, KV * D), o=(NH * D, H),
                                                          gate=(H, I), up=(H, I), down=(I, H)).items()}
        # act-order: q / k / v share one permutation, gate / up another (same inputs -> same Hessian diagonal), o / down own
        gi = {n: None for n in parts}
        if act_order:
            gq, gg = order_of(H), order_of(H)
            gi = dict(q=gq, k=gq, v=gq, gate=gg, up=gg, o=order_of(NH * D), down=order_of(I))
            if not isinstance(act_order, bool):  # a subset of {"qkv", "o", "gate_up", "down"} (diagnostics)
                keep = {"qkv": ("q", "k", "v"), "o": ("o",), "gate_up": ("gate", "up"), "down": ("down",)}
                live = {n for grp in act_order for n in keep[grp]}
                gi = {n: (g if n in live else None) for n, g in gi.items()}
        for n, (q, s, z) in parts.items():
            ly[n] = cpu_pack(q, s, z, gi[n])
        cat = lambda i: np.concatenate([parts["q"][i], parts["k"][i], parts["v"][i]], 1)  # noqa: E731
        qkv = gpu_pack(cat(0), cat(1), cat(2) if asym else None, gi["q"])
        o = gpu_pack(*parts["o"], gi["o"])
        tt = lambda a: torch.from_numpy(a)  # noqa: E731
        gu = gpu_pack(fuse_gate_up(tt(parts["gate"][0]), tt(parts["up"][0])).numpy(),
                      fuse_gate_up(tt(parts["gate"][1]), tt(parts["up"][1])).numpy(),
                      fuse_gate_up(tt(parts["gate"][2]), tt(parts["up"][2])).numpy() if asym else None, gi["gate"])
        down = gpu_pack(*parts["down"], gi["down"])
        ly["ln1"] = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        ly["ln2"] = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        eng.set_layer(l, qkv, o, gu, down, torch.from_numpy(ly["ln1"]), torch.from_numpy(ly["ln2"]))
        layers.append(ly)
    embed = torch.from_numpy(rng.standard_normal((cfg["vocab"], H)).astype(np.float32)).half()
    lm = torch.from_numpy((rng.standard_normal((cfg["vocab"], H)) * 0.1).astype(np.float32)).half()
    norm = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    eng.set_head(embed, torch.from_numpy(norm), lm)
    oracle = orc.LlamaOracle(cfg, embed.float().numpy(), layers, norm, lm.float().numpy())
    return eng, oracle, cfg


@pytest.mark.parametrize("group,asym,scale_dtype", [(128, False, "fp16"), (32, True, "fp32"), (-1, False, "bf16")])
def test_engine_logits_vs_oracle(group, asym, scale_dtype):
    eng, oracle, cfg = _tiny(group, asym, scale_dtype)
    prompt = [3, 17, 200, 5, 99, 42]
    worst = 0.0
    for i, t in enumerate(prompt):
        eng.token.fill_(t)
        eng.pos.fill_(i)
        eng.step(greedy=False)
        got = eng.logits.cpu().numpy()
        ref = oracle.forward_token(t, i)
        err = np.abs(got - ref).max()
        worst = max(worst, err / np.abs(ref).max())
        assert err <= 2e-3 * np.abs(ref).max() + 1e-4, (i, err, np.abs(ref).max())
        assert int(got.argmax()) == int(ref.argmax())
    print("worst relative logit error", worst)


def test_engine_fp32_activation_path_still_matches(monkeypatch):
    """The engine hands activations between its kernels as XQ limb blocks by default (csrc/woq_xq.h); WOQ_ENGINE_XQ=0
    selects the fp32-activation kernels (the path tensor-parallel ranks and the module use). Both against the oracle,
    and against each other within the fixed-point step (per-16 vs per-slice exponents: not bit-identical)."""
    monkeypatch.setenv("WOQ_ENGINE_XQ", "0")
    eng0, oracle, cfg = _tiny(32, True, "fp16", seed=2)
    monkeypatch.delenv("WOQ_ENGINE_XQ")
    eng1, _, _ = _tiny(32, True, "fp16", seed=2)
    for i, t in enumerate([3, 17, 200, 5]):
        outs = []
        for eng in (eng0, eng1):
            eng.token.fill_(t)
            eng.pos.fill_(i)
            eng.step(greedy=False)
            outs.append(eng.logits.cpu().numpy())
        ref = oracle.forward_token(t, i)
        for got in outs:
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4
        assert np.abs(outs[0] - outs[1]).max() <= 1e-4 * np.abs(ref).max() + 1e-5


def test_engine_graph_replay_matches_eager():
    """hipGraph replay of the captured step must reproduce the eager greedy token chain bit for bit."""
    eng, oracle, cfg = _tiny(128, False, "fp16", seed=1)
    eager = eng.generate([5, 9, 2], 8)
    eng2, _, _ = _tiny(128, False, "fp16", seed=1)
    for i, t in enumerate([5, 9, 2]):
        eng2.token.fill_(t)
        eng2.pos.fill_(i)
        eng2.step(greedy=(i == 2))
    eng2.capture(greedy=True)
    toks = [int(eng2.token.item())]
    for _ in range(7):
        eng2.replay(1)
        toks.append(int(eng2.token.item()))
    assert toks == eager
    # and the oracle agrees on the greedy chain
    oracle.reset()
    ref = []
    seq = [5, 9, 2]
    for i, t in enumerate(seq):
        lg = oracle.forward_token(t, i)
    for j in range(8):
        nxt = int(lg.argmax())
        ref.append(nxt)
        lg = oracle.forward_token(nxt, len(seq) + j)
    assert ref == eager


def test_eager_bursts_equal_graph_replays():
    """runtime/engine.py launch modes: a burst of n decode steps issued eagerly through ONE native call
    (`woq_engine_steps`, the product default since round 4) against n replays of the captured hipGraph and against n
    separate `step()` calls: identical greedy token chains and bit-identical final logits; `replay()` follows the
    engine's launch mode."""
    outs = {}
    for mode in ("graph", "eager", "steps"):
        eng, _, _ = _tiny(128, False, "fp16", seed=1)
        eng.launch = "eager" if mode == "eager" else "graph"
        for i, t in enumerate([5, 9, 2]):
            eng.token.fill_(t)
            eng.pos.fill_(i)
            eng.step(greedy=(i == 2))
        first = int(eng.token.item())
        if mode == "steps":
            for _ in range(24):
                eng.step(greedy=True)
        else:
            eng.prepare_decode(greedy=True)
            assert eng.captured == (mode == "graph")
            eng.replay(5)   # graph mode: five single-step launches
            eng.replay(19)  # two launches of the 8-step graph + three single steps (round 5: woq_engine_replay)
        torch.cuda.synchronize()
        outs[mode] = ([first] + eng.token_log()[3:27].tolist(), eng.logits.clone())
        assert eng.status() == 0
    assert outs["graph"][0] == outs["eager"][0] == outs["steps"][0]
    assert torch.equal(outs["graph"][1], outs["eager"][1]) and torch.equal(outs["graph"][1], outs["steps"][1])


# ---- prompt pass (woq_engine_prefill) ---------------------------------------------------------------------------
# Stated tolerance: the prefill linears contract fp16 operands (11-bit significands; the reference's own
# reduced-precision cores use bf16, 8 bits) and keep q / k / v / attention / MLP activations in fp16 between kernels,
# fp32 residual stream and accumulation. Bound on the last-position logits against the fp32 oracle on the same
# (q, scale, zp): max|diff| <= 1e-2 * max|logit_oracle| + 1e-3 (measured: a few 1e-4 relative), greedy token equal.
PF_TOL = 1e-2


def _oracle_prompt(oracle, prompt):
    oracle.reset()
    lg = None
    for i, t in enumerate(prompt):
        lg = oracle.forward_token(int(t), i)
    return lg


@pytest.mark.parametrize("group,asym,scale_dtype,head_dim,T", [(128, False, "fp16", 64, 7), (32, True, "fp32", 64, 150),
                                                              (128, False, "fp16", 128, 200),
                                                              (-1, False, "bf16", 128, 130)])
def test_prefill_logits_vs_oracle(group, asym, scale_dtype, head_dim, T):
    eng, oracle, cfg = _tiny(group, asym, scale_dtype, seed=3, max_ctx=256, head_dim=head_dim)
    rng = np.random.default_rng(5)
    prompt = rng.integers(0, cfg["vocab"], T).tolist()
    got = eng.prefill(prompt, greedy=True)[0].cpu().numpy()
    ref = _oracle_prompt(oracle, prompt)
    err = np.abs(got - ref).max()
    print("prefill relative logit error", err / np.abs(ref).max())
    assert err <= PF_TOL * np.abs(ref).max() + 1e-3
    assert int(eng.token.item()) == int(ref.argmax()) and int(eng.pos.item()) == T
    assert np.array_equal(eng.logits.cpu().numpy(), got)
    # decode continues on the cache the prompt pass wrote
    nxt = int(ref.argmax())
    for j in range(3):
        eng.step(greedy=True)
        ref = oracle.forward_token(nxt, T + j)
        g = eng.logits.cpu().numpy()
        assert np.abs(g - ref).max() <= PF_TOL * np.abs(ref).max() + 1e-3
        nxt = int(ref.argmax())
        assert int(eng.token.item()) == nxt


def test_prefill_chunked_and_batched_match():
    """Chunked prefill (3 calls with growing start_pos) and a 3-sequence batch reproduce the one-shot result;
    a bf16 KV cache stays inside the same bound."""
    T = 150
    rng = np.random.default_rng(6)
    prompts = rng.integers(0, 384, (3, T))
    eng, oracle, cfg = _tiny(128, False, "fp16", seed=4, max_ctx=256, max_batch=3)
    full = eng.prefill(prompts[0].tolist())[0].cpu().numpy().copy()
    ref0 = _oracle_prompt(oracle, prompts[0])
    assert np.abs(full - ref0).max() <= PF_TOL * np.abs(ref0).max() + 1e-3
    # chunks of 64 + 64 + 22: same cache contents up to fp16 rounding of identical arithmetic -> tight agreement
    for s0 in (0, 64, 128):
        lg = eng.prefill(prompts[0][s0:s0 + 64].tolist(), start_pos=s0)
    chunked = lg[0].cpu().numpy()
    assert np.abs(chunked - full).max() <= 2e-3 * np.abs(full).max() + 1e-4
    batch = eng.prefill(prompts).cpu().numpy()
    assert np.abs(batch[0] - full).max() <= 2e-3 * np.abs(full).max() + 1e-4
    for s in (1, 2):
        ref = _oracle_prompt(oracle, prompts[s])
        assert np.abs(batch[s] - ref).max() <= PF_TOL * np.abs(ref).max() + 1e-3
        assert int(batch[s].argmax()) == int(ref.argmax())
    engb, oracleb, _ = _tiny(128, False, "fp16", seed=4, max_ctx=256, kv_dtype=torch.bfloat16)
    gb = engb.prefill(prompts[0].tolist())[0].cpu().numpy()
    assert np.abs(gb - ref0).max() <= 3e-2 * np.abs(ref0).max() + 1e-3  # bf16 K/V: 2^-9 per cached element


def test_prefill_matches_token_by_token_decode():
    """The prompt pass and the batch-1 decode path fill the same cache: greedy continuation agrees."""
    prompt = [5, 9, 2, 77, 300, 41, 8, 19, 250, 3]
    eng, _, _ = _tiny(128, False, "fp16", seed=1, max_ctx=64)
    a = eng.generate(prompt, 8)
    eng2, _, _ = _tiny(128, False, "fp16", seed=1, max_ctx=64)
    for i, t in enumerate(prompt):
        eng2.token.fill_(t)
        eng2.pos.fill_(i)
        eng2.step(greedy=(i == len(prompt) - 1))
    b = [int(eng2.token.item())]
    for _ in range(7):
        eng2.step(greedy=True)
        b.append(int(eng2.token.item()))
    assert a == b


def test_prefill_argument_checks():
    eng, _, _ = _tiny(128, False, "fp16", seed=1, max_ctx=64)
    with pytest.raises(RuntimeError, match="max_ctx"):
        eng.prefill(list(range(60)), start_pos=10)
    with pytest.raises(RuntimeError, match="max_batch"):
        eng.prefill(np.zeros((2, 4), dtype=np.int64))


def test_fp8_kv_cache_prefill_and_decode():
    """kv_dtype fp8_e4m3 (BASELINE configs[4]): the cache holds OCP e4m3fn bytes of the same K / V the fp16 engine
    holds (3-bit significand: |diff| <= 2^-4 |x| + smallest subnormal step), no NaN codes, and both the prompt pass
    and the decode step read it back: logits stay within 6e-2 * max|logit| of the fp16-cache engine on a
    150-token prompt and 3 decode steps."""
    T = 150
    rng = np.random.default_rng(9)
    prompt = rng.integers(0, 384, T).tolist()
    e16, _, cfg = _tiny(128, False, "fp16", seed=7, max_ctx=256, head_dim=128)
    e8, _, _ = _tiny(128, False, "fp16", seed=7, max_ctx=256, head_dim=128, kv_dtype=torch.float8_e4m3fn)
    l16 = e16.prefill(prompt)[0].cpu().numpy().copy()
    l8 = e8.prefill(prompt)[0].cpu().numpy().copy()
    for which in ("k", "v"):
        # layer 0 only: deeper layers see hidden states that already differ through layer 0's fp8 attention
        c16 = e16.kv_cache(which)[0, :1, :T].float().cpu().numpy()
        raw = e8.kv_cache(which)[0, :1, :T]
        bits = raw.view(torch.uint8).cpu().numpy()
        assert ((bits & 0x7f) != 0x7f).all()  # e4m3fn NaN codes never appear
        c8 = raw.float().cpu().numpy()
        assert (np.abs(c8 - c16) <= 2.0 ** -4 * np.abs(c16) + 2.0 ** -9).all()
    assert np.abs(l8 - l16).max() <= 6e-2 * np.abs(l16).max()
    for _ in range(3):
        e16.token.copy_(e8.token)  # same continuation token on both engines
        e16.step(greedy=True)
        e8.step(greedy=True)
        a, b = e16.logits.cpu().numpy(), e8.logits.cpu().numpy()
        assert np.abs(a - b).max() <= 6e-2 * np.abs(a).max()
        e16.token.copy_(e8.token)
    # the decode step's own append wrote e4m3 too
    k8 = e8.kv_cache("k")[0, :1, T:T + 3].float().cpu().numpy()
    k16 = e16.kv_cache("k")[0, :1, T:T + 3].float().cpu().numpy()
    assert (np.abs(k8 - k16) <= 2.0 ** -4 * np.abs(k16) + 2.0 ** -9).all()


@pytest.mark.parametrize("head_dim,splits,grouped,hidden", [(64, 3, False, 256), (128, 5, False, 256),
                                                            (128, 5, True, 256), (128, 3, True, 512),
                                                            (128, 5, True, 1024)])
def test_decode_attention_context_slices_vs_oracle(head_dim, splits, grouped, hidden):
    """Long-context decode attention (context slices + combine launch), forced on at a small context: token-by-token
    decode over 200 positions — empty slices, slice boundaries, ragged last sub-tiles, the new position in the last
    slice — against the fp32 oracle at the decode tolerance, and identical greedy tokens under graph replay. Both
    sliced forms: one workgroup per (query head, slice), and the grouped-query matrix-core form (one kv head with
    2 / 4 / 8 query heads: hidden 256 / 512 / 1024 at head_dim 128)."""
    eng, oracle, cfg = _tiny(128, False, "fp16", seed=2, max_ctx=256, head_dim=head_dim, attn_splits=splits,
                             hidden=hidden, attn_grouped=grouped)
    assert cfg["kv_heads"] == 128 // head_dim and cfg["heads"] == hidden // head_dim
    rng = np.random.default_rng(3)
    toks = rng.integers(0, cfg["vocab"], 200).tolist()
    for i, t in enumerate(toks):
        eng.token.fill_(t)
        eng.pos.fill_(i)
        eng.step(greedy=False)
        ref = oracle.forward_token(t, i)
        if i in (0, 1, 63, 64, 65, 127, 128, 129, 191, 192, 199):
            got = eng.logits.cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4, i
    eng.token.fill_(int(ref.argmax()))
    eng.pos.fill_(200)
    eng.capture(greedy=True)
    nxt = int(ref.argmax())
    for j in range(4):
        eng.replay(1)
        ref = oracle.forward_token(nxt, 200 + j)
        nxt = int(ref.argmax())
        assert int(eng.token.item()) == nxt


@pytest.mark.parametrize("hidden,chunk,splits", [(512, 32, 5), (512, 64, 3), (1024, 64, 4), (256, 96, 3)])
def test_grouped_attention_fixed_chunk_slices_vs_oracle(hidden, chunk, splits, monkeypatch):
    """Round 4: the grouped-query sliced decode attention with POSITION-INDEPENDENT slices (slice s owns the absolute
    positions [s * chunk, (s + 1) * chunk), K / V requested before the device-side position is read) and the merge done
    by the last slice workgroup to finish (csrc/woq_attn_merge.h) instead of a combine launch. Token by token over 200
    positions with small chunks: slices wholly beyond the position (empty partials), the position crossing slice
    boundaries, the new token's row landing in every slice in turn, and the last slice's overflow (splits * chunk <
    200) — against the fp32 oracle at the decode tolerance; then graph replays (the arrival counters must come back to
    zero by themselves) with identical greedy tokens."""
    monkeypatch.setenv("WOQ_ATTN_FOLD", "1")  # read at engine creation (the in-launch merge is opt-in: measured slower)
    eng, oracle, cfg = _tiny(128, False, "fp16", seed=5, max_ctx=256, head_dim=128, attn_splits=splits, hidden=hidden,
                             attn_grouped=True)
    monkeypatch.delenv("WOQ_ATTN_FOLD")
    eng.set_attn_chunk(chunk)
    rng = np.random.default_rng(8)
    toks = rng.integers(0, cfg["vocab"], 200).tolist()
    check = {0, 1, chunk - 1, chunk, chunk + 1, 2 * chunk - 1, 2 * chunk, splits * chunk - 1, splits * chunk,
             splits * chunk + 1, 150, 199}
    for i, t in enumerate(toks):
        eng.token.fill_(t)
        eng.pos.fill_(i)
        eng.step(greedy=False)
        ref = oracle.forward_token(t, i)
        if i in check:
            got = eng.logits.cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4, i
    eng.token.fill_(int(ref.argmax()))
    eng.pos.fill_(200)
    eng.capture(greedy=True)
    nxt = int(ref.argmax())
    for j in range(6):
        eng.replay(1)
        ref = oracle.forward_token(nxt, 200 + j)
        nxt = int(ref.argmax())
        assert int(eng.token.item()) == nxt
    assert eng.status() == 0


@pytest.mark.parametrize("hidden,splits,kv_dtype", [(512, 4, torch.float16), (1024, 7, torch.float16), (256, 32, torch.float16),
                                                    (512, 5, torch.float8_e4m3fn)])
def test_grouped_slices_merging_among_themselves_equal_the_combine_launch(hidden, splits, kv_dtype, monkeypatch):
    """Round 6: the grouped-query matrix-core slices publish tagged partial granules and finalise the group's output
    blocks themselves (csrc/woq_attn_merge.h, all-to-all merge: block B of the group's REP x 8 goes to slice B mod ns)
    instead of ending in a combine launch (WOQ_GROUPED_A2A=0, read at engine creation) — same sums in the same order:
    logits and greedy tokens BIT-IDENTICAL over 150 decode steps (empty slices, more slices than blocks: 32 > 16,
    fewer: 4 < 32, a slice count that does not divide the blocks: 7), then as graph replays; status clean."""
    engs = []
    for a2a in ("0", "1"):
        monkeypatch.setenv("WOQ_GROUPED_A2A", a2a)
        engs.append(_tiny(128, False, "fp16", seed=6, max_ctx=256, head_dim=128, attn_splits=splits, hidden=hidden,
                          attn_grouped=True, kv_dtype=kv_dtype)[0])
    cfg_vocab = 384
    rng = np.random.default_rng(9)
    for i, t in enumerate(rng.integers(0, cfg_vocab, 150).tolist()):
        outs = []
        for e in engs:
            e.token.fill_(t)
            e.pos.fill_(i)
            e.step(greedy=True)
            outs.append((e.logits.clone(), int(e.token.item())))
        assert outs[0][1] == outs[1][1], i
        assert torch.equal(outs[0][0], outs[1][0]), i
    toks = []
    for e in engs:
        e.pos.fill_(150)
        e.capture(greedy=True)
        e.replay_graph(10)
        torch.cuda.synchronize()
        toks.append((e.logits.clone(), e.token_log()[150:160].clone()))
        assert e.status() == 0
    assert torch.equal(toks[0][0], toks[1][0]) and torch.equal(toks[0][1], toks[1][1])


@pytest.mark.parametrize("grouped,hidden,head_dim", [(False, 256, 64), (False, 256, 128), (True, 512, 128)])
def test_slice_merge_by_last_workgroup_equals_the_combine_launch(grouped, hidden, head_dim, monkeypatch):
    """The two ways of merging context-slice partials — the last slice workgroup of a head (default) and the separate
    combine launch (WOQ_ATTN_FOLD=0, the A/B twin) — run the same sums in the same order: logits equal to 1e-6 of the
    largest over 150 decode steps, greedy tokens identical."""
    monkeypatch.setenv("WOQ_ATTN_FOLD", "0")
    e0, _, cfg = _tiny(128, False, "fp16", seed=6, max_ctx=256, head_dim=head_dim, attn_splits=4, hidden=hidden,
                       attn_grouped=grouped)
    monkeypatch.setenv("WOQ_ATTN_FOLD", "1")
    e1, _, _ = _tiny(128, False, "fp16", seed=6, max_ctx=256, head_dim=head_dim, attn_splits=4, hidden=hidden,
                     attn_grouped=grouped)
    rng = np.random.default_rng(9)
    for i, t in enumerate(rng.integers(0, cfg["vocab"], 150).tolist()):
        outs = []
        for e in (e0, e1):
            e.token.fill_(t)
            e.pos.fill_(i)
            e.step(greedy=True)
            outs.append((e.logits.clone(), int(e.token.item())))
        assert outs[0][1] == outs[1][1], i
        assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-6 * outs[0][0].abs().max().item() + 1e-7, i


def test_tp_seam_world_size_one_rccl():
    """The tensor-parallel plumbing on a real device with a one-rank RCCL group: TPDecoder.prefill (native prompt pass
    + all-reduce callback into torch.distributed + vocab all-gather) and TPDecoder.step (host-driven sub-blocks +
    all-reduces) must reproduce the single-GPU engine — a sum over one rank is the identity, so any difference is a
    plumbing bug (wrong buffer, count, stream or ordering)."""
    import os

    import torch.distributed as dist

    from intel_extension_for_transformers_amd.runtime.tp import TPDecoder

    import socket

    with socket.socket() as sock:  # a free port: the fixed default may be taken on a shared box
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng, oracle, cfg = _tiny(128, False, "fp16", seed=5, max_ctx=256)
        ref_eng, _, _ = _tiny(128, False, "fp16", seed=5, max_ctx=256)
        prompt = np.random.default_rng(1).integers(0, cfg["vocab"], 70).tolist()
        want = ref_eng.prefill(prompt, greedy=True)[0].cpu().numpy().copy()
        tp = TPDecoder(eng, cfg["vocab"])
        got = tp.prefill(prompt)[0].cpu().numpy()
        assert np.array_equal(got, want)
        # prefill leaves the engine where the single-GPU prefill(greedy=True) does: next token in place, pos = T
        assert int(eng.token.item()) == int(ref_eng.token.item()) == int(want.argmax())
        assert int(eng.pos.item()) == int(ref_eng.pos.item()) == len(prompt)
        for _ in range(3):
            lg = tp.step(greedy=True).cpu().numpy()
            ref_eng.step(greedy=True)
            assert np.array_equal(lg, ref_eng.logits.cpu().numpy())
            assert int(eng.token.item()) == int(ref_eng.token.item())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("head_dim,splits,grouped", [(64, 0, False), (128, 3, False), (128, 3, True)])
def test_sliding_window_attention_vs_oracle(head_dim, splits, grouped):
    """HF Mistral `sliding_window` (a query sees the last W positions, itself included): prompt pass over 150 tokens
    (window 40: whole tiles below the window are skipped, both tile edges masked), then decode steps — one workgroup
    per head and the sliced form — and token-by-token decode from an empty cache, against the fp32 oracle with the
    same window."""
    W = 40
    eng, oracle, cfg = _tiny(128, False, "fp16", seed=8, max_ctx=256, head_dim=head_dim, attn_splits=splits, window=W,
                             attn_grouped=grouped)
    rng = np.random.default_rng(4)
    prompt = rng.integers(0, cfg["vocab"], 150).tolist()
    got = eng.prefill(prompt, greedy=True)[0].cpu().numpy()
    ref = _oracle_prompt(oracle, prompt)
    assert np.abs(got - ref).max() <= PF_TOL * np.abs(ref).max() + 1e-3
    nxt = int(ref.argmax())
    for j in range(4):
        eng.step(greedy=True)
        ref = oracle.forward_token(nxt, 150 + j)
        assert np.abs(eng.logits.cpu().numpy() - ref).max() <= PF_TOL * np.abs(ref).max() + 1e-3
        nxt = int(ref.argmax())
    eng2, oracle2, _ = _tiny(128, False, "fp16", seed=8, max_ctx=256, head_dim=head_dim, attn_splits=splits, window=W,
                             attn_grouped=grouped)
    for i, t in enumerate(prompt[:90]):
        eng2.token.fill_(t)
        eng2.pos.fill_(i)
        eng2.step(greedy=False)
        r = oracle2.forward_token(t, i)
        if i in (0, 38, 39, 40, 41, 64, 89):
            assert np.abs(eng2.logits.cpu().numpy() - r).max() <= 2e-3 * np.abs(r).max() + 1e-4, i


@pytest.mark.parametrize("grouped", [False, True])
def test_sliced_attention_grouped_queries_rep4_fp8(grouped):
    """Sliced decode attention under grouped queries (4 q heads / 1 kv head) with an fp8 KV cache: 5 slices,
    token-by-token over 150 positions against the engine's own one-workgroup-per-head form. Per-query-head slices
    differ from it by summation order only, but a last-bit difference in layer 0's attention output survives the
    per-16-value fixed-point conversion of the XQ hand-off (csrc/woq_xq.h) and can flip an fp8 rounding of layer 1's
    cache (a 6 % step of that element): both forms get the decode tolerance; the grouped matrix-core form also rounds
    the probabilities to fp16 for the P V product."""
    kw = dict(seed=9, max_ctx=256, head_dim=128, hidden=512, kv_dtype=torch.float8_e4m3fn)
    a, _, cfg = _tiny(128, False, "fp16", attn_splits=5, attn_grouped=grouped, **kw)
    b, _, _ = _tiny(128, False, "fp16", attn_splits=1, **kw)
    assert cfg["heads"] == 4 and cfg["kv_heads"] == 1
    rtol, atol = 2e-3, 1e-4
    rng = np.random.default_rng(2)
    for i, t in enumerate(rng.integers(0, cfg["vocab"], 150).tolist()):
        for e in (a, b):
            e.token.fill_(t)
            e.pos.fill_(i)
            e.step(greedy=False)
        if i in (0, 1, 15, 16, 63, 64, 65, 127, 128, 149):
            la, lb = a.logits.cpu().numpy(), b.logits.cpu().numpy()
            assert np.abs(la - lb).max() <= rtol * np.abs(lb).max() + atol, i


@pytest.mark.parametrize("rep,kv_dtype", [(4, torch.float16), (8, torch.float8_e4m3fn), (2, torch.float16)])
def test_grouped_attention_long_slices_match_per_head_slices(rep, kv_dtype):
    """The grouped-query form at a context where every wave walks several sub-tiles (1000 cached positions in 2
    slices of 512: 16 sub-tiles per slice, 4 per wave — both register sets, both refills, a ragged last sub-tile),
    against the per-query-head slices on an identical cache (same chunked prompt pass on both engines). Explicit
    tokens at explicit positions, so the two engines never branch on a near-tie."""
    kw = dict(seed=11, max_ctx=1280, head_dim=128, hidden=128 * rep, kv_dtype=kv_dtype, attn_splits=2)
    a, _, cfg = _tiny(128, False, "fp16", attn_grouped=True, **kw)
    b, _, _ = _tiny(128, False, "fp16", **kw)
    assert cfg["heads"] == rep and cfg["kv_heads"] == 1
    rng = np.random.default_rng(12)
    prompt = rng.integers(0, cfg["vocab"], 1000)
    for e in (a, b):
        for s0 in range(0, 1000, 250):
            e.prefill(prompt[s0:s0 + 250].tolist(), start_pos=s0)
    assert torch.equal(a.kv_cache("k")[0, :, :1000].float(), b.kv_cache("k")[0, :, :1000].float())
    for j, t in enumerate(rng.integers(0, cfg["vocab"], 4).tolist()):
        for e in (a, b):
            e.token.fill_(t)
            e.pos.fill_(1000 + j)
            e.step(greedy=False)
        la, lb = a.logits.cpu().numpy(), b.logits.cpu().numpy()
        assert np.abs(la - lb).max() <= 2e-3 * np.abs(lb).max() + 1e-4, j


def test_step_beyond_max_ctx_is_flagged_and_stays_in_bounds():
    """The position lives on the device (greedy chaining advances it there), so woq_engine_step / _replay cannot check
    it: a step that starts at or beyond max_ctx runs at max_ctx - 1 instead and raises bit 1 of the sticky status —
    no out-of-bounds KV-cache / RoPE-table / token-log access (round-2 ADVICE)."""
    eng, _, cfg = _tiny(128, False, "fp16", max_ctx=16)
    assert eng.status() == 0
    eng.token.fill_(5)
    eng.pos.fill_(14)
    eng.capture(greedy=True)
    eng.replay(1)   # position 14 -> 15: fine
    assert eng.status() == 0 and int(eng.pos.item()) == 15
    eng.replay(6)   # 15 is the last slot; the following steps start at 16 = max_ctx
    torch.cuda.synchronize()
    assert eng.status() & 2
    assert int(eng.pos.item()) <= 16
    assert torch.isfinite(eng.logits).all()


@pytest.mark.gpu
@pytest.mark.parametrize("asym,group", [(False, 128), (True, 32)])
def test_short_prompt_split_k_prompt_pass_equals_decode_steps(asym, group):
    """A 40-token prompt on a mid-size decoder (hidden 1024, inter 512, 8 heads): every linear of the prompt pass is a
    one-row-block GEMM of 8-24 workgroups, which run as K slices (split-K, csrc/woq_gemm_f16.hip) — with the SiLU * mul
    (gate / up) and the residual (o / down) applied after the slices are summed. Against the SAME engine's decode path
    (GEMV kernels, another code path entirely) fed the same 40 tokens one by one: last-position logits within 1e-2 of
    the largest (fp16-operand GEMMs vs fp32-class GEMVs), same greedy token."""
    from intel_extension_for_transformers_amd.runtime import WoqDecoderEngine, synth_llama_weights

    H, I, NH, D, L, V = 1024, 512, 8, 128, 2, 1000
    eng = WoqDecoderEngine(H, I, NH, NH, D, L, V, max_ctx=64)
    synth_llama_weights(eng, H, I, NH, NH, D, L, V, group=group, sym=not asym, scale_dtype="fp16")
    g = torch.Generator().manual_seed(5)
    toks = torch.randint(0, V, (40,), generator=g).tolist()
    for i, t in enumerate(toks):
        eng.token.fill_(int(t))
        eng.pos.fill_(i)
        eng.step(greedy=False)
    torch.cuda.synchronize()
    ref = eng.logits.clone()
    got = eng.prefill(toks, greedy=False)[0]
    torch.cuda.synchronize()
    assert (got - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    assert int(got.argmax()) == int(ref.argmax())


@pytest.mark.parametrize("group,asym,scale_dtype", [(128, False, "fp16"), (32, True, "fp32"), (64, True, "bf16")])
def test_engine_act_order_layers_vs_oracle(group, asym, scale_dtype):
    """GPTQ act-order (desc_act) layers INSIDE the fused engine (round 5): every projection carries a
    g_idx; the decode step runs the fp32-activation tile GEMVs with the act-order gather (fused RMSNorm prologue,
    residual / SiLU*mul epilogues unchanged), the prompt pass the MFMA GEMM whose pack pass gathers — logits and greedy
    tokens against the oracle decoder, whose linears apply `index_select(x, 1, g_idx)` (autograd/functions.py:41-63),
    eager bursts == graph replays."""
    eng, oracle, cfg = _tiny(group, asym, scale_dtype, seed=11, act_order=True)
    assert not eng.uses_xq() and not eng.uses_fused_attn()
    prompt = [3, 17, 200, 5, 99, 42, 7]
    for i, t in enumerate(prompt):
        eng.token.fill_(t)
        eng.pos.fill_(i)
        eng.step(greedy=False)
        got, ref = eng.logits.cpu().numpy(), oracle.forward_token(t, i)
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4, (i, np.abs(got - ref).max())
        assert int(got.argmax()) == int(ref.argmax())
    eng.launch = "eager"
    eager = eng.generate(prompt, 10)
    eng.launch = "graph"
    graph = eng.generate(prompt, 10)
    assert eager == graph and eng.status() == 0
    oracle.reset()
    logits = oracle.forward_prompt(prompt)
    want = []
    for j in range(10):
        want.append(int(np.argmax(logits)))
        logits = oracle.forward_token(want[-1], len(prompt) + j)
    assert eager == want
    oracle.reset()
    got, ref = eng.prefill(prompt)[0].cpu().numpy(), oracle.forward_prompt(prompt)
    assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-4


@pytest.mark.parametrize("weight_dtype,group,scale_dtype,compute_dtype",
                         [("nf4", 128, "fp16", "fp32"), ("nf4", 32, "fp32", "fp32"), ("nf4", 128, "fp16", "bf16"),
                          ("fp4_e2m1", 128, "bf16", "fp32"), ("fp4_e2m1_bnb", -1, "fp32", "fp32")])
def test_engine_table_weight_types_vs_oracle(weight_dtype, group, scale_dtype, compute_dtype):
    """nf4 / fp4 layers in the fused engine (round 4; reference strings bestla_weightonly_dispatcher.hpp:62-70): the
    decode step's XQ GEMVs with the digit-plane unpack (csrc/woq_gemv_common.h LutArgs) — eager steps, graph replays
    and the fp32-activation kernels — and the prompt pass on the MFMA GEMM over a pre-dequantised fragment image (fused
    RMSNorm and SiLU*mul epilogues), all against the oracle decoder on the same codes and scales. nf4 blobs packed for
    compute bf16 decode with two digit planes (table held to 2^-16 of its largest entry) instead of three."""
    import os

    eng, oracle, cfg = _tiny(group, False, scale_dtype, seed=7, weight_dtype=weight_dtype,
                             compute_dtype=compute_dtype)
    assert not eng.uses_fused_attn()  # the fused qkv + attention launch stays int4-only
    prompt = [3, 17, 200, 5, 99, 42, 7]
    for i, t in enumerate(prompt):
        eng.token.fill_(t)
        eng.pos.fill_(i)
        eng.step(greedy=False)
        got, ref = eng.logits.cpu().numpy(), oracle.forward_token(t, i)
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4, (i, np.abs(got - ref).max())
        assert int(got.argmax()) == int(ref.argmax())
    # greedy chain: eager bursts == graph replays, and the oracle's tokens
    eng.launch = "eager"
    eager = eng.generate(prompt, 10)
    eng.launch = "graph"
    graph = eng.generate(prompt, 10)
    assert eager == graph
    oracle.reset()
    logits = oracle.forward_prompt(prompt)
    want = []
    for j in range(10):
        want.append(int(np.argmax(logits)))
        logits = oracle.forward_token(want[-1], len(prompt) + j)
    assert eager == want
    # prompt pass logits (fragment-image GEMM: one fp16 product per operand pair, like the int4 prompt pass)
    oracle.reset()
    got, ref = eng.prefill(prompt)[0].cpu().numpy(), oracle.forward_prompt(prompt)
    assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-4
    # the fp32-activation kernels (WOQ_ENGINE_XQ=0: woq_gemv_i8.hip with the same unpack)
    os.environ["WOQ_ENGINE_XQ"] = "0"
    try:
        eng0, oracle0, _ = _tiny(group, False, scale_dtype, seed=7, weight_dtype=weight_dtype,
                                 compute_dtype=compute_dtype)
    finally:
        del os.environ["WOQ_ENGINE_XQ"]
    for i, t in enumerate(prompt[:4]):
        eng0.token.fill_(t)
        eng0.pos.fill_(i)
        eng0.step(greedy=False)
        got, ref = eng0.logits.cpu().numpy(), oracle0.forward_token(t, i)
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4


@pytest.mark.parametrize("weight_dtype,group,scale_dtype,hidden", [("fp8_e4m3", 128, "fp16", 256), ("fp8_e4m3", 32, "fp32", 256),
                                                                 ("fp8_e5m2", -1, "bf16", 256), ("fp8_e4m3", 128, "fp32", 512)])
def test_engine_fp8_weight_layers_vs_oracle(weight_dtype, group, scale_dtype, hidden):
    """Round 6 (reference weight strings bestla_weightonly_dispatcher.hpp:62-72): fp8_e4m3 / fp8_e5m2
    layers INSIDE the decode engine. set_layer takes the composite containers (HI / LO nibble planes), the decode step
    runs the fp8 matrix-core GEMVs with RMSNorm / residual fused (csrc/woq_gemv_fp8.hip F8Fused) and a SiLU * mul pairing
    launch, the prompt pass the MFMA GEMM over the pre-dequantised fragment image. Token-by-token decode, a prompt pass
    and graph replays against the fp32 oracle decoder on the SAME code bytes and scales: decode logits within
    2e-3 * max|logit| + 1e-4 (the engine tolerance), prompt-pass logits within the fp16-operand bound, greedy tokens equal.
    Per-128 groups (the fp8 kernel's fast form), per-32 groups (its per-32-scale form), one group per column with e5m2."""
    eng, oracle, cfg = _tiny(group, False, scale_dtype, seed=13, max_ctx=96, head_dim=64, hidden=hidden,
                             weight_dtype=weight_dtype)
    assert not eng.uses_xq() and not eng.uses_fused_attn()
    rng = np.random.default_rng(17)
    toks = rng.integers(0, cfg["vocab"], 24).tolist()
    for i, t in enumerate(toks):
        eng.token.fill_(t)
        eng.pos.fill_(i)
        eng.step(greedy=True)
        ref = oracle.forward_token(t, i)
        got = eng.logits.cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4, i
        assert int(eng.token.item()) == int(ref.argmax()), i
    # graph replays continue the same sequence
    nxt = int(ref.argmax())
    eng.capture(greedy=True)
    for j in range(4):
        eng.replay(1)
        ref = oracle.forward_token(nxt, 24 + j)
        nxt = int(ref.argmax())
        assert int(eng.token.item()) == nxt
    # prompt pass over fresh tokens (fp16-operand GEMMs over the dequantised fragment image)
    oracle.reset()
    prompt = rng.integers(0, cfg["vocab"], 40).tolist()
    got = eng.prefill(prompt, greedy=True)[0].cpu().numpy()
    ref = oracle.forward_prompt(prompt)
    assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max() + 1e-3
    assert int(got.argmax()) == int(ref.argmax())
    eng.step(greedy=True)
    ref = oracle.forward_token(int(ref.argmax()), 40)
    assert np.abs(eng.logits.cpu().numpy() - ref).max() <= 1e-2 * np.abs(ref).max() + 1e-3
    assert eng.status() == 0

