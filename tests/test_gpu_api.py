"""GPU checks of the ITREX-compatible Python API: QuantizedLinearQBits, convert_to_quantized_model,
AutoModelForCausalLM.from_pretrained / save_low_bit / load_low_bit, and the fused decode-engine post-pass.

Models are tiny random-init HF architectures (no checkpoints, no network), saved to a temp dir so that the
`from_pretrained(path, quantization_config=...)` route of the reference's own tests
(tests/CI/test_weight_only.py:159-209) is the one exercised. Reference outputs: the SAME HF architecture in fp32
torch with every quantised linear's weight replaced by its dequantised weight — the reference's self-consistency
criterion (qbits_ut/test_weightonly.py:51-88), tolerance stated per assert.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import woq_oracle as orc

pytestmark = pytest.mark.gpu


def _tiny_llama(kv_heads=4):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=kv_heads, vocab_size=320, max_position_embeddings=256, rms_norm_eps=1e-5,
                      tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).float().eval()


def _tiny_gpt2():
    from transformers import GPT2Config, GPT2LMHeadModel

    torch.manual_seed(0)
    return GPT2LMHeadModel(GPT2Config(n_embd=128, n_layer=2, n_head=4, vocab_size=300, n_positions=128)).float().eval()


def _dequantised_twin(qmodel, fp_model):
    """fp32 copy of the architecture whose converted linears carry the dequantised weights of `qmodel`."""
    from intel_extension_for_transformers_amd import qbits
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    twin = copy.deepcopy(fp_model).cuda()
    qmods = dict(qmodel.named_modules())
    n = 0
    for name, mod in twin.named_modules():
        qm = qmods.get(name)
        if isinstance(qm, QuantizedLinearQBits):
            k, nn_ = qm.in_features, qm.out_features
            deq = torch.empty(k, nn_, dtype=torch.float32, device="cuda")
            qbits.dequantize_packed_weight(qm.weight.data, deq, False, "fp32", "int4_clip", qm.scale_dtype)
            with torch.no_grad():
                mod.weight.copy_(deq if type(mod).__name__ == "Conv1D" else deq.t())
            n += 1
    assert n > 0
    return twin


def test_tiny_linear_int4_module():
    """reference tests/CI/test_weight_only.py:117-137 (there with int8): M(32 -> 2), with and without bias."""
    from intel_extension_for_transformers_amd import qbits
    from intel_extension_for_transformers_amd.transformers import RtnConfig
    from intel_extension_for_transformers_amd.transformers.llm.quantization.utils import convert_to_quantized_model
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    class M(torch.nn.Module):
        def __init__(self, with_bias):
            super().__init__()
            self.linear = torch.nn.Linear(32, 2, bias=with_bias)

        def forward(self, x):
            return self.linear(x)

    torch.manual_seed(0)
    raw = torch.rand(2, 32)
    packed = qbits.quantize_to_packed_weight(raw.cuda(), True, 32, "fp32", "int4_clip", "fp32", False)
    revert = torch.zeros(2, 32, device="cuda")
    qbits.dequantize_packed_weight(packed, revert, True, "fp32", "int4_clip", "fp32")
    for bias in (True, False):
        model = M(bias)
        with torch.no_grad():
            model.linear.weight = torch.nn.Parameter(revert.cpu())  # exactly representable -> re-quantises exactly
        act = torch.rand(1, 32)
        ref = model(act)
        cfg = RtnConfig(bits=4, weight_dtype="int4", group_size=32)
        cfg.post_init_hip()
        convert_to_quantized_model(model, cfg, device="cuda")
        assert isinstance(model.linear, QuantizedLinearQBits)
        out = model(act.cuda()).cpu()
        assert torch.allclose(ref, out, rtol=0.01, atol=1e-5)  # the reference's own tolerance


@pytest.mark.parametrize("method,sym,desc_act", [("rtn", True, False), ("rtn", False, False), ("gptq", False, True),
                                                 ("gptq", True, False)])
def test_set_weights_bias_and_recover(method, sym, desc_act):
    """Checkpoint-side tensors (unsigned int weight, scales, zeros +1-unbiased, GPTQ g_idx) -> blob -> forward vs
    the oracle on the same tensors (modules.py:195-262), then recover_qparms returns them exactly (:297-392)."""
    from intel_extension_for_transformers_amd.transformers import GPTQConfig, RtnConfig
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    K, N, g = 256, 48, 32
    rng = np.random.default_rng(5)
    w_u = rng.integers(0, 16, (K, N)).astype(np.int8)
    s = ((rng.random((K // g, N), dtype=np.float32) + 0.5) * 0.02)
    z_u = rng.integers(1, 16, (K // g, N)).astype(np.int8)
    gidx = rng.permutation(np.arange(K, dtype=np.int32) // g).astype(np.int32)
    cfg = (GPTQConfig(bits=4, group_size=g, sym=sym, desc_act=desc_act) if method == "gptq"
           else RtnConfig(bits=4, group_size=g, sym=sym))
    cfg.post_init_hip()
    m = QuantizedLinearQBits(K, N, True, compute_dtype="fp32", weight_dtype="int4_clip", bits=4, scale_dtype="fp32",
                             blocksize=g, scheme=cfg.scheme)
    bias = torch.from_numpy(rng.random(N, dtype=np.float32))
    m.set_weights_bias(torch.from_numpy(w_u), torch.from_numpy(s), torch.from_numpy(z_u), torch.from_numpy(gidx), cfg,
                       bias)
    # oracle on the same tensors: signed domain, rows regrouped by convert_idx when act-order is on
    q = w_u.astype(np.int16) - 8
    zp = None if sym else (z_u.astype(np.int16) - 8).astype(np.int8)
    shuf = None
    if desc_act:
        shuf = orc.convert_idx(gidx, K, g)
        q = q[shuf]
    blob = orc.repack(q.astype(np.int8), s, zp, shuf, g)
    x = (rng.random((3, K), dtype=np.float32) - 0.4)
    ref = orc.woq_linear(x, blob, bias.numpy())
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-6
    # recover: the checkpoint-side tensors come back exactly, in the checkpoint's own row order and with the RAW
    # g_idx (group id per row) — not the blob's regrouped rows / converted shuffle (reference recover_idx,
    # recover_int_weight, modules.py:299-326) — so feeding them to set_weights_bias again is the identity
    iw, sc, zz, gi = m.recover_qparms_kn()
    assert np.array_equal(iw.cpu().numpy(), w_u)
    assert np.array_equal(sc.cpu().numpy(), s)
    if sym:
        assert zz is None
    else:
        assert np.array_equal(zz.cpu().numpy(), z_u)
    if desc_act:
        assert np.array_equal(gi.cpu().numpy(), gidx)
    else:
        assert gi is None
    m2 = QuantizedLinearQBits(K, N, True, compute_dtype="fp32", weight_dtype="int4_clip", bits=4, scale_dtype="fp32",
                              blocksize=g, scheme=cfg.scheme)
    m2.set_weights_bias(iw, sc, zz, gi if gi is not None else torch.empty(0, dtype=torch.int32), cfg, bias)
    assert torch.equal(m2.weight.data, m.weight.data)  # byte-identical blob
    # the reference's 12-tuple view of the same thing (modules.py:378-392), in ITS orientation
    t = m.recover_qparms()
    assert len(t) == 12 and t[0] == g and t[1] == K and t[2] == N and t[3] == desc_act and t[6] == 4 and t[9] == (not sym)
    assert t[5] == "int4_clip" and t[7] == torch.float32
    assert np.array_equal(t[8].cpu().numpy(), s.T)
    # symmetric: the reference's int_weight is the SIGNED round(w / scale) (no zero point to add, modules.py:356-372);
    # asymmetric: unsigned (the +8 sits on the zero points, :349-352)
    assert np.array_equal(t[11].cpu().numpy(), (w_u.astype(np.int16) - 8).T if sym else w_u.T)
    assert (t[10] is None) if sym else np.array_equal(t[10].cpu().numpy(), z_u.T)


@pytest.mark.parametrize("group,sym,scale_dtype", [(128, True, "fp16"), (32, False, "fp32")])
def test_from_pretrained_llama_logits_generate_save_load(tmp_path, group, sym, scale_dtype):
    """reference tests/CI/test_weight_only.py:159-209: from_pretrained(..., quantization_config) swaps the linears,
    generate runs, save -> load round trip reproduces the outputs."""
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    fp = _tiny_llama()
    src = tmp_path / "fp"
    fp.save_pretrained(str(src))
    qmodel = AutoModelForCausalLM.from_pretrained(str(src), quantization_config=RtnConfig(
        bits=4, group_size=group, sym=sym, scale_dtype=scale_dtype), use_neural_speed=False)
    lin = [m for m in qmodel.modules() if isinstance(m, torch.nn.Linear)]
    assert sum(isinstance(m, QuantizedLinearQBits) for m in lin) == 2 * 7  # 7 projections per layer
    assert not isinstance(qmodel.lm_head, QuantizedLinearQBits)            # llm_int8_skip_modules (config.py:836-837)
    assert qmodel.quantization_config.group_size == group
    ids = torch.tensor([[5, 17, 200, 3, 77, 140, 9, 31]], device="cuda")
    with torch.no_grad():
        logits = qmodel(ids).logits.float()
        twin = _dequantised_twin(qmodel, fp)
        ref = twin(ids).logits.float()
    # same math, fp32 both sides; differences are summation order only
    assert (logits - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-5
    out = qmodel.generate(ids, max_new_tokens=8, do_sample=False)
    ref_out = twin.generate(ids, max_new_tokens=8, do_sample=False)
    assert torch.equal(out, ref_out)
    # config.token_latency -> (ids, latency_list), one entry per generated token (reference greedy_search.py:148-150,
    # 374-377,408-409; asserted by tests/CI/test_weight_only.py:173-183)
    qmodel.config.token_latency = True
    res = qmodel.generate(ids, max_new_tokens=8, do_sample=False)
    qmodel.config.token_latency = False
    assert len(res) == 2 and isinstance(res[1], list) and torch.equal(res[0], out)
    assert len(res[1]) == out.shape[1] - ids.shape[1] and all(t > 0 for t in res[1])
    dst = tmp_path / "woq"
    qmodel.save_pretrained(str(dst))
    reloaded = AutoModelForCausalLM.from_pretrained(str(dst))
    with torch.no_grad():
        again = reloaded(ids).logits.float()
    assert torch.equal(again, logits)  # identical integers, scales and kernels -> identical bits


def test_from_pretrained_gpt2_conv1d(tmp_path):
    """BASELINE configs[0] plumbing: GPT-2's projections are Conv1D, weight already [K, N]."""
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    fp = _tiny_gpt2()
    src = tmp_path / "gpt2"
    fp.save_pretrained(str(src))
    qmodel = AutoModelForCausalLM.from_pretrained(str(src), load_in_4bit=True)
    nq = sum(isinstance(m, QuantizedLinearQBits) for m in qmodel.modules())
    assert nq == 2 * 4  # c_attn, c_proj, mlp.c_fc, mlp.c_proj per block
    ids = torch.tensor([[1, 2, 3, 4, 5, 6]], device="cuda")
    with torch.no_grad():
        logits = qmodel(ids).logits.float()
        ref = _dequantised_twin(qmodel, fp)(ids).logits.float()
    assert (logits - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-5
    out = qmodel.generate(ids, max_new_tokens=6, do_sample=False, pad_token_id=0)
    assert out.shape == (1, 12)


@pytest.mark.parametrize("kv_heads", [4, 2])
def test_optimize_transformers_engine_matches_hf_path(tmp_path, kv_heads):
    """The fused decode engine built from the quantised HF model reproduces that model's greedy decode
    (the HF path runs torch attention / norm kernels around the same int4 linears)."""
    from intel_extension_for_transformers_amd.runtime.engine import optimize_transformers
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig

    fp = _tiny_llama(kv_heads)
    src = tmp_path / "fp"
    fp.save_pretrained(str(src))
    qmodel = AutoModelForCausalLM.from_pretrained(str(src), quantization_config=RtnConfig(bits=4, group_size=128,
                                                                                          scale_dtype="fp16"))
    eng = optimize_transformers(qmodel, max_ctx=64, kv_dtype=torch.float16)
    prompt = [5, 17, 200, 3, 77]
    ids = torch.tensor([prompt], device="cuda")
    with torch.no_grad():
        ref_logits = qmodel(ids).logits[0, -1].float()
        ref_out = qmodel._woq_hf_generate(ids, max_new_tokens=6, do_sample=False)[0, len(prompt):].tolist()
    for i, t in enumerate(prompt):
        eng.token.fill_(t)
        eng.pos.fill_(i)
        eng.step(greedy=False)
    got = eng.logits.float()
    # fp16 KV cache + fp16 lm_head on the engine side vs fp32 torch: 2e-3 relative on logits
    assert (got - ref_logits).abs().max().item() <= 2e-3 * ref_logits.abs().max().item() + 1e-4
    assert eng.generate(prompt, 6) == ref_out


def _tiny_tokenizer(path, vocab_size):
    """Word-level tokenizer over w0..wN (no network): enough for build_chatbot's tokenizer.from_pretrained."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for i in range(3, vocab_size):
        vocab["w%d" % i] = i
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>")
    fast.save_pretrained(str(path))


def test_neural_chat_build_chatbot_predict_and_stream(tmp_path):
    """SURVEY §8(f)3 / north_star: PipelineConfig(optimization_config=RtnConfig) -> build_chatbot -> predict /
    predict_stream. Greedy requests ride the fused engine (prompt pass + graph-replayed decode); the streamed pieces
    concatenate to predict()'s text and match the HF module path's greedy continuation; sampling requests (the reference's
    default) ride the engine with the device sampler (anything else takes the model.generate + TextIteratorStreamer route);
    return_stats appends the reference's stats table."""
    from intel_extension_for_transformers_amd.neural_chat import GenerationConfig, PipelineConfig, build_chatbot
    from intel_extension_for_transformers_amd.transformers import RtnConfig

    d = tmp_path / "tiny-llama-chat"
    fp = _tiny_llama()
    fp.generation_config.eos_token_id = None  # random-init model: never stop early
    fp.save_pretrained(str(d))
    _tiny_tokenizer(d, fp.config.vocab_size)
    bot = build_chatbot(PipelineConfig(model_name_or_path=str(d), device="cuda",
                                       optimization_config=RtnConfig(bits=4, group_size=128, scale_dtype="fp16")))
    assert type(bot).__name__ == "LlamaModel" and bot.engine is not None
    q = "w5 w17 w200 w3 w77"
    cfg = GenerationConfig(max_new_tokens=8, do_sample=False, repetition_penalty=1.0)
    text = bot.predict(q, config=cfg)
    stream, link = bot.predict_stream(q, config=cfg)  # (generator, link) like the reference
    assert link == []
    pieces = list(stream)
    assert "".join(pieces) == text and len(text.split()) == 8
    # the module path (HF generate over the QuantizedLinearQBits model) gives the same greedy continuation
    bot_engine, bot.engine = bot.engine, None
    bot.model._woq_engine_off = True  # and keep model.generate itself on the HF loop
    try:
        assert bot.predict(q, config=cfg) == text
    finally:
        bot.engine = bot_engine
        bot.model._woq_engine_off = False
    # sampling (the reference's DEFAULT GenerationConfig: do_sample, temperature 0.1, top_k 40, top_p 0.75, repetition
    # penalty 1.1) rides the engine too since round 4 (device sampler); stats block in the v2 table format
    scfg = GenerationConfig(max_new_tokens=5, return_stats=True)
    assert scfg.do_sample and scfg.repetition_penalty == 1.1

    def no_hf(*a, **k):
        raise AssertionError("a default NeuralChat request must not fall back to HF's loop")

    hf_stream, bot._hf_stream = bot._hf_stream, no_hf
    try:
        out = list(bot.predict_stream(q, config=scfg)[0])
    finally:
        bot._hf_stream = hf_stream
    joined = "".join(out)
    assert "| Key" in joined and "msecond_per_token" in joined and "input_token_len" in joined
    assert len(joined.split("| Key")[0].split()) == 5
    # temperature 0.1 over a random-init model is all but greedy: with top_k = 1 it IS the penalised greedy chain
    g1 = bot.predict(q, config=GenerationConfig(max_new_tokens=6, top_k=1))
    g2 = bot.predict(q, config=GenerationConfig(max_new_tokens=6, do_sample=False))
    assert g1 == g2


def test_from_pretrained_bits8_int8_weights(tmp_path):
    """`bits in {4, 8}` (reference utils/config.py:277-372): RtnConfig(bits=8) / load_in_8bit -> int8 weights on the
    module path; logits against the dequantised fp32 twin; save_low_bit -> load_low_bit round trip."""
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig

    fp = _tiny_llama()
    src = tmp_path / "fp"
    fp.save_pretrained(str(src))
    qmodel = AutoModelForCausalLM.from_pretrained(str(src), quantization_config=RtnConfig(bits=8, group_size=64))
    assert qmodel.quantization_config.weight_dtype == "int8"
    twin = _dequantised_twin8(qmodel, fp)
    ids = torch.tensor([[5, 17, 200, 3, 77, 12]], device="cuda")
    with torch.no_grad():
        a = qmodel(ids).logits.float()
        b = twin(ids).logits.float()
    assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-5
    # 8-bit RTN of N(0, 0.02) weights is close to the fp model itself
    with torch.no_grad():
        c = fp.cuda()(ids).logits.float()
    assert (a - c).abs().max().item() <= 3e-2 * c.abs().max().item()
    out = tmp_path / "q8"
    qmodel.save_pretrained(str(out))
    again = AutoModelForCausalLM.from_pretrained(str(out))
    with torch.no_grad():
        d = again(ids).logits.float()
    assert (a - d).abs().max().item() <= 1e-5 * a.abs().max().item() + 1e-6
    m2 = AutoModelForCausalLM.from_pretrained(str(src), load_in_8bit=True)
    assert m2.quantization_config.bits == 8


def _dequantised_twin8(qmodel, fp_model):
    from intel_extension_for_transformers_amd import qbits
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    twin = copy.deepcopy(fp_model).cuda()
    qmods = dict(qmodel.named_modules())
    for name, mod in twin.named_modules():
        qm = qmods.get(name)
        if isinstance(qm, QuantizedLinearQBits):
            deq = torch.empty(qm.in_features, qm.out_features, dtype=torch.float32, device="cuda")
            qbits.dequantize_packed_weight(qm.weight.data, deq, False, "fp32", "int8", qm.scale_dtype)
            with torch.no_grad():
                mod.weight.copy_(deq.t())
    return twin


def test_model_generate_is_engine_backed_for_greedy(tmp_path):
    """`model.generate` of a Llama-class int4 model rides the fused engine for the requests it covers (one sequence,
    greedy) and HF's loop for the rest; both give the same greedy continuation; eos stops the engine path; a
    streamer receives the prompt and every new token."""
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig

    fp = _tiny_llama()
    fp.generation_config.eos_token_id = None
    src = tmp_path / "fp"
    fp.save_pretrained(str(src))
    qmodel = AutoModelForCausalLM.from_pretrained(str(src), quantization_config=RtnConfig(bits=4, group_size=128,
                                                                                          scale_dtype="fp16"))
    ids = torch.tensor([[5, 17, 200, 3, 77, 12, 9]], device="cuda")
    hf = qmodel._woq_hf_generate(ids, max_new_tokens=8, do_sample=False, pad_token_id=0)
    assert not hasattr(qmodel, "woq_engine")
    out = qmodel.generate(ids, max_new_tokens=8, do_sample=False, pad_token_id=0)
    assert hasattr(qmodel, "woq_engine") and torch.equal(out, hf)
    # eos: stop right after the first token the model itself would emit
    first = int(hf[0, ids.shape[1]])
    short = qmodel.generate(ids, max_new_tokens=8, do_sample=False, eos_token_id=first)
    assert short.shape[1] == ids.shape[1] + 1 and int(short[0, -1]) == first

    class Collect:
        def __init__(self):
            self.items, self.ended = [], False

        def put(self, v):
            self.items.append(v.reshape(-1).tolist())

        def end(self):
            self.ended = True

    st = Collect()
    qmodel.generate(ids, max_new_tokens=4, do_sample=False, streamer=st)
    assert st.ended and st.items[0] == ids[0].tolist() and sum(st.items[1:], []) == hf[0, 7:11].tolist()
    # a sampling request is not the engine's: it takes HF's loop and still works
    torch.manual_seed(0)
    samp = qmodel.generate(ids, max_new_tokens=4, do_sample=True, top_k=5, pad_token_id=0)
    assert samp.shape == (1, ids.shape[1] + 4)


def test_generate_reads_the_engine_status_and_retries_or_raises():
    """the fused qkv + attention launch reports a timed-out in-launch hand-off (and a position
    clamp) through the engine's sticky status word only. `generate` reads it once after the last burst: hand-off
    time-out -> the engine drops to separate launches and the request is re-run (same tokens as a clean run); with a
    streamer, or for any other bit, RuntimeError — never a silent wrong result. The status itself: a step beyond
    max_ctx sets bit 1, `clear_status` resets it."""
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig

    fp = _tiny_llama()
    q = AutoModelForCausalLM.from_pretrained(copy.deepcopy(fp), quantization_config=RtnConfig(
        bits=4, group_size=32, compute_dtype="fp32", scale_dtype="fp32"), device_map="cuda")
    ids = torch.tensor([[5, 9, 33, 2, 71]], device="cuda")
    clean = q.generate(ids, max_new_tokens=12, do_sample=False)
    eng = q.woq_engine
    assert eng.status() == 0
    real = eng.status
    seen = {"n": 0}

    def flaky():
        seen["n"] += 1
        return 1 if seen["n"] == 1 else real()

    eng.status = flaky
    again = q.generate(ids, max_new_tokens=12, do_sample=False)
    assert torch.equal(again, clean) and seen["n"] == 2 and not eng.uses_fused_attn()
    eng.status = lambda: 2
    with pytest.raises(RuntimeError, match="status 2"):
        q.generate(ids, max_new_tokens=12, do_sample=False)
    eng.status = real
    # the real word: a step at max_ctx is clamped and says so; clear_status resets it
    eng.token.fill_(3)
    eng.pos.fill_(eng.cfg.max_ctx + 4)
    eng.step(greedy=True)
    assert eng.status() == 2
    eng.clear_status()
    assert eng.status() == 0
    assert torch.equal(q.generate(ids, max_new_tokens=12, do_sample=False), clean)


def _write_hf_gptq_checkpoint(fp, d, group, sym, desc_act, shards=2, seed=0, shared_perm=True):
    """A Hugging Face GPTQ checkpoint directory (AutoGPTQ tensor names / packing) synthesised from a tiny fp model with
    the oracle's RTN per (act-ordered) group: qweight int32 [K/8, N] in ORIGINAL row order, g_idx [K], qzeros storing
    zp - 1, fp16 scales. Returns {linear name: dequantised weight [N, K]} for the fp32 twin."""
    import json
    import os

    from safetensors.torch import save_file

    from intel_extension_for_transformers_amd.transformers.llm.quantization.utils import pack_weight

    rng = np.random.default_rng(seed)
    tensors, deq, perms = {}, {}, {}
    for name, mod in fp.named_modules():
        if not isinstance(mod, torch.nn.Linear) or name == "lm_head":
            continue
        w = mod.weight.detach().float().numpy().T.copy()  # [K, N]
        K, N = w.shape
        # a real GPTQ run orders the rows by the Hessian diagonal of the layer's INPUT, so projections that share their
        # input (q / k / v; gate / up) come out with the same permutation: `shared_perm` writes that; without it every
        # linear gets its own (a checkpoint the fused engine must refuse and the module path must still serve)
        leaf = name.rsplit(".", 1)[-1]
        key = {"k_proj": "q_proj", "v_proj": "q_proj", "up_proj": "gate_proj"}.get(leaf, leaf) if shared_perm else leaf
        pkey = name.rsplit(".", 1)[0] + "." + key
        if pkey not in perms:
            perms[pkey] = rng.permutation(K) if desc_act else np.arange(K)
        perm = perms[pkey]
        q_p, s, z = orc.rtn_quantize(w[perm], False, group, not sym)  # groups are contiguous in the permuted order
        s = s.astype(np.float16).astype(np.float32)
        q = np.empty_like(q_p)
        q[perm] = q_p
        g_idx = np.empty(K, np.int32)
        g_idx[perm] = np.arange(K, dtype=np.int32) // group
        zz = np.zeros_like(s) if z is None else z.astype(np.float32)
        deq[name] = ((q.astype(np.float32) - zz[g_idx]) * s[g_idx]).T.copy()
        uz = torch.from_numpy((z.astype(np.int16) + 8) if z is not None else np.full(s.shape, 8, np.int16))
        qweight, sc16, qzeros = pack_weight(torch.from_numpy(q.astype(np.int16) + 8), torch.from_numpy(s), uz, bits=4)
        tensors[name + ".qweight"], tensors[name + ".scales"] = qweight.contiguous(), sc16.contiguous()
        tensors[name + ".qzeros"], tensors[name + ".g_idx"] = qzeros.contiguous(), torch.from_numpy(g_idx)
    for k, v in fp.state_dict().items():
        owner = k.rsplit(".", 1)[0]
        if owner + ".qweight" not in tensors:
            tensors[k] = v.detach().clone().contiguous()
    os.makedirs(d, exist_ok=True)
    keys = sorted(tensors)
    weight_map = {}
    for i in range(shards):
        part = {k: tensors[k] for k in keys[i::shards]}
        fname = "model-%05d-of-%05d.safetensors" % (i + 1, shards)
        save_file(part, os.path.join(d, fname), metadata={"format": "pt"})
        weight_map.update({k: fname for k in part})
    with open(os.path.join(d, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": weight_map}, f)
    cfg = fp.config.to_dict()
    cfg["quantization_config"] = {"quant_method": "gptq", "bits": 4, "group_size": group, "sym": sym,
                                  "desc_act": desc_act, "static_groups": False, "damp_percent": 0.1}
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    return deq


@pytest.mark.parametrize("sym,desc_act,shared_perm", [(True, False, True), (False, True, True), (False, True, False)])
def test_from_pretrained_hf_gptq_checkpoint(tmp_path, sym, desc_act, shared_perm):
    """a Hugging Face GPTQ checkpoint directory (sharded safetensors, AutoGPTQ names,
    zp - 1 zeros, act-order g_idx) loads straight to the GPU layout; logits match the fp32 twin carrying the
    checkpoint's own dequantised weights; generate runs on the fused engine — act-order checkpoints included (round 5:
    the decode GEMV gathers its activations by the stored shuffle), as long as q / k / v and gate / up share their
    permutation the way a GPTQ run leaves them; a checkpoint with a different permutation per linear is refused by the
    engine and served by the module path, same tokens."""
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM

    fp = _tiny_llama()
    fp.generation_config.eos_token_id = None
    d = tmp_path / "tiny-llama-gptq"
    deq = _write_hf_gptq_checkpoint(fp, str(d), 128, sym, desc_act, shared_perm=shared_perm)
    model = AutoModelForCausalLM.from_pretrained(str(d))
    assert model.quantization_config.quant_method in ("gptq", getattr(model.quantization_config.quant_method, "value", None))
    twin = copy.deepcopy(fp).cuda()
    with torch.no_grad():
        for name, mod in twin.named_modules():
            if name in deq:
                mod.weight.copy_(torch.from_numpy(deq[name]))
    ids = torch.tensor([[5, 17, 200, 3, 77, 140, 9, 31]], device="cuda")
    with torch.no_grad():
        a = model(ids).logits.float()
        b = twin(ids).logits.float()
    assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-5
    out = model.generate(ids, max_new_tokens=6, do_sample=False, pad_token_id=0)
    ref = twin.generate(ids, max_new_tokens=6, do_sample=False, pad_token_id=0)
    assert torch.equal(out, ref)
    assert hasattr(model, "woq_engine") == (not desc_act or shared_perm)
    if desc_act and shared_perm:  # the engine really decoded those tokens, on the fp32-activation kernels with the gather
        assert not model.woq_engine.uses_xq() and model.woq_engine.status() == 0


def test_from_pretrained_hf_awq_checkpoint(tmp_path):
    """BASELINE configs[2] is "AWQ-style" (asym, small groups): a Hugging Face AWQ checkpoint directory (AutoAWQ "GEMM"
    tensors packed along N, zero points as they are) loads straight to the GPU layout and reproduces its own
    dequantised fp32 twin; the fused engine serves its greedy generate."""
    import json
    import os

    from safetensors.torch import save_file

    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM
    from intel_extension_for_transformers_amd.transformers.llm.quantization.utils import pack_awq_gemm

    fp = _tiny_llama()
    fp.generation_config.eos_token_id = None
    d = str(tmp_path / "tiny-llama-awq")
    os.makedirs(d)
    group, tensors, deq = 32, {}, {}
    for name, mod in fp.named_modules():
        if not isinstance(mod, torch.nn.Linear) or name == "lm_head":
            continue
        w = mod.weight.detach().float().numpy().T.copy()  # [K, N]
        q, s, z = orc.rtn_quantize(w, False, group, True)
        s = s.astype(np.float16).astype(np.float32)
        K = w.shape[0]
        rows = np.arange(K) // group
        deq[name] = ((q.astype(np.float32) - z.astype(np.float32)[rows]) * s[rows]).T.copy()
        qw, qz = pack_awq_gemm(torch.from_numpy(q.astype(np.int16) + 8), torch.from_numpy(z.astype(np.int16) + 8))
        tensors[name + ".qweight"], tensors[name + ".qzeros"] = qw.contiguous(), qz.contiguous()
        tensors[name + ".scales"] = torch.from_numpy(s).half().contiguous()
    for k, v in fp.state_dict().items():
        if k.rsplit(".", 1)[0] + ".qweight" not in tensors:
            tensors[k] = v.detach().clone().contiguous()
    save_file(tensors, os.path.join(d, "model.safetensors"), metadata={"format": "pt"})
    cfg = fp.config.to_dict()
    cfg["quantization_config"] = {"quant_method": "awq", "bits": 4, "group_size": group, "zero_point": True,
                                  "version": "gemm"}
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    model = AutoModelForCausalLM.from_pretrained(d)
    twin = copy.deepcopy(fp).cuda()
    with torch.no_grad():
        for name, mod in twin.named_modules():
            if name in deq:
                mod.weight.copy_(torch.from_numpy(deq[name]))
    ids = torch.tensor([[5, 17, 200, 3, 77, 140, 9, 31]], device="cuda")
    with torch.no_grad():
        a, b = model(ids).logits.float(), twin(ids).logits.float()
    assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-5
    out = model.generate(ids, max_new_tokens=6, do_sample=False, pad_token_id=0)
    assert torch.equal(out, twin.generate(ids, max_new_tokens=6, do_sample=False, pad_token_id=0))
    assert hasattr(model, "woq_engine")


@pytest.mark.parametrize("wname,bits,sname", [("nf4", 4, "fp32"), ("fp4_e2m1", 4, "fp32"), ("fp8_e4m3", 8, "fp32"),
                                              ("fp8_e5m2", 8, "fp8_e8m0")])
def test_from_pretrained_table_weight_dtype(tmp_path, wname, bits, sname):
    """RtnConfig(weight_dtype="nf4" | "fp4_e2m1" | "fp8_e4m3" | "fp8_e5m2") (reference docs/weightonlyquant.md dtype
    table; fp8 weights take fp32 or power-of-two fp8_e8m0 scales): quantised on the device, logits of the module path
    against the fp32 twin carrying the dequantised weights; greedy generate — the fused engine for the 4-bit table types
    (round 4: codes read back from the modules' blobs, runtime/engine.py _code_parts), the module path for fp8."""
    from intel_extension_for_transformers_amd import qbits
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    fp = _tiny_llama()
    fp.generation_config.eos_token_id = None
    src = tmp_path / "fp"
    fp.save_pretrained(str(src))
    qmodel = AutoModelForCausalLM.from_pretrained(str(src), quantization_config=RtnConfig(bits=bits, group_size=64,
                                                                                          weight_dtype=wname,
                                                                                          scale_dtype=sname))
    assert qmodel.quantization_config.weight_dtype == wname and qmodel.quantization_config.scale_dtype == sname
    twin = copy.deepcopy(fp).cuda()
    qmods = dict(qmodel.named_modules())
    with torch.no_grad():
        for name, mod in twin.named_modules():
            qm = qmods.get(name)
            if isinstance(qm, QuantizedLinearQBits):
                deq = torch.empty(qm.in_features, qm.out_features, dtype=torch.float32, device="cuda")
                qbits.dequantize_packed_weight(qm.weight.data, deq, False, "fp32", wname, qm.scale_dtype)
                mod.weight.copy_(deq.t())
    ids = torch.tensor([[5, 17, 200, 3, 77, 140, 9, 31]], device="cuda")
    with torch.no_grad():
        a, b = qmodel(ids).logits.float(), twin(ids).logits.float()
    assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-5
    out = qmodel.generate(ids, max_new_tokens=4, do_sample=False, pad_token_id=0)
    assert torch.equal(out, twin.generate(ids, max_new_tokens=4, do_sample=False, pad_token_id=0))
    assert hasattr(qmodel, "woq_engine") == (bits == 4)
    if bits == 4:  # the engine's fused qkv / gate-up blobs hold exactly the modules' weights
        eng = qmodel.woq_engine
        at = qmodel.model.layers[0].self_attn
        H = at.q_proj.in_features
        parts = []
        for m in (at.q_proj, at.k_proj, at.v_proj):
            d = torch.empty(H, m.out_features, dtype=torch.float32, device="cuda")
            qbits.dequantize_packed_weight(m.weight.data, d, False, "fp32", wname, sname)
            parts.append(d)
        want = torch.cat(parts, 1)
        got = torch.empty_like(want)
        qbits.dequantize_packed_weight(eng.layer_tensors[0]["qkv"], got, False, "fp32", wname, sname)
        assert torch.equal(got, want)
    with pytest.raises(ValueError, match="asym"):
        RtnConfig(bits=bits, weight_dtype=wname, sym=False).post_init_hip()


def test_quantized_linear_vllm_return_convention(monkeypatch):
    """backend=use_vllm (set by the reference's from_pretrained(use_vllm=True), modeling_auto.py:364-480): forward
    returns vLLM's (output, output_bias) pair so the module can stand in for vLLM's parallel linears
    (reference modules.py:166-167)."""
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import QuantizedLinearQBits

    torch.manual_seed(4)
    lin = QuantizedLinearQBits(128, 32, True, compute_dtype="fp32", weight_dtype="int4_clip", scale_dtype="fp32",
                               blocksize=64, scheme="sym")
    lin.set_fp_weights_bias(torch.randn(32, 128) * 0.05, torch.randn(32))
    x = torch.randn(3, 128, device="cuda")
    monkeypatch.delenv("backend", raising=False)
    plain = lin(x)
    assert isinstance(plain, torch.Tensor) and plain.shape == (3, 32)
    monkeypatch.setenv("backend", "use_vllm")
    out, out_bias = lin(x)
    assert out_bias is None and torch.equal(out, plain)


def test_set_woq_workspace_makes_scratch_calls_capturable():
    """qbits.set_woq_workspace (reference qbits.cpp:142-144): with a caller-owned workspace the calls that need
    scratch — 16-bit activations at small M, int8 weights, the M > 8 GEMM — allocate nothing, so they can be captured
    into a graph (a hipMallocAsync inside a capture fails), and replaying the graph reproduces the eager results."""
    from intel_extension_for_transformers_amd import qbits

    K, N = 512, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn(N, K, generator=g, device="cuda") * 0.05
    blob4 = qbits.quantize_to_packed_weight(w, True, 128, "fp32", "int4_clip", "fp32", False)
    blob8 = qbits.quantize_to_packed_weight(w, True, 128, "fp32", "int8", "fp32", False)
    e = torch.empty(0)
    cases = [(blob4, torch.randn(2, K, generator=g, device="cuda").bfloat16(), "int4_clip"),   # 16-bit rows (read natively, no scratch)
             (blob8, torch.randn(3, K, generator=g, device="cuda"), "int8"),                  # int8 composite
             (blob4, torch.randn(40, K, generator=g, device="cuda"), "int4_clip")]            # MFMA GEMM pack pass
    eager = []
    for blob, x, wt in cases:
        out = torch.empty(x.shape[0], N, device="cuda", dtype=x.dtype)
        qbits.woq_linear(x, blob, e, out, "fp32", wt, "fp32", False)
        eager.append(out.clone())
    ws = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    qbits.set_woq_workspace(ws)
    try:
        outs = [torch.empty_like(o) for o in eager]
        for (blob, x, wt), out in zip(cases, outs):  # warm (lazy kernel attributes) outside the capture
            qbits.woq_linear(x, blob, e, out, "fp32", wt, "fp32", False)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for (blob, x, wt), out in zip(cases, outs):
                qbits.woq_linear(x, blob, e, out, "fp32", wt, "fp32", False)
        for out in outs:
            out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for out, ref in zip(outs, eager):
            assert torch.equal(out, ref)
    finally:
        qbits.set_woq_workspace(None)


@pytest.mark.parametrize("layout", ["sharded", "bin"])
def test_desc_act_checkpoint_save_load_round_trip(tmp_path, layout):
    """An act-order GPTQ model (desc_act: rows regrouped at load, activations gathered by the kernels) survives
    save_pretrained -> from_pretrained: recover_qparms hands back the RAW g_idx and the checkpoint's own row order
    (reference recover_idx / recover_int_weight, modules.py:299-326), so the reload regroups once, not twice. The saved
    directory uses HF's file layout (sharded safetensors + index, or pytorch_model.bin) like the reference's."""
    import os

    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM

    fp = _tiny_llama()
    src = tmp_path / "gptq-src"
    _write_hf_gptq_checkpoint(fp, str(src), 128, False, True)
    model = AutoModelForCausalLM.from_pretrained(str(src))
    ids = torch.tensor([[5, 17, 200, 3, 77, 140, 9, 31]], device="cuda")
    with torch.no_grad():
        want = model(ids).logits.float()
    out = tmp_path / "resaved"
    if layout == "sharded":
        model.save_pretrained(str(out), max_shard_size="200KB")
        assert os.path.isfile(out / "model.safetensors.index.json")
        assert len([f for f in os.listdir(out) if f.startswith("model-") and f.endswith(".safetensors")]) > 1
    else:
        model.save_pretrained(str(out), safe_serialization=False)
        assert os.path.isfile(out / "pytorch_model.bin")
    assert os.path.isfile(out / "quantize_config.json")
    again = AutoModelForCausalLM.from_pretrained(str(out))
    with torch.no_grad():
        got = again(ids).logits.float()
    assert torch.equal(got, want)  # same integers, same scales, same shuffle -> the same blobs
    for (na, ma), (nb, mb) in zip(model.named_modules(), again.named_modules()):
        if hasattr(ma, "recover_qparms_kn"):
            assert torch.equal(ma.weight.data, mb.weight.data), na


def test_generate_sampling_and_repetition_penalty_ride_the_engine():
    """The reference's NeuralChat default request samples (do_sample=True, temperature 0.1, top_k 40, top_p 0.75,
    repetition_penalty 1.1 — neural_chat/config.py:400-409); such requests now run on the fused engine with the next
    token chosen on the device (runtime.engine.DeviceSampler) instead of HF's loop over the per-linear modules.
    Deterministic cases against HF's own generate over the quantised modules: repetition penalty without sampling
    (token for token), sampling with top_k = 1 (== greedy with the penalty); sampling proper: reproducible under a seed,
    every token inside the top-k set HF's processors leave at that step, eos honoured, streamer fed."""
    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig

    fp = _tiny_llama()
    fp.generation_config.eos_token_id = None
    q = AutoModelForCausalLM.from_pretrained(copy.deepcopy(fp), quantization_config=RtnConfig(
        bits=4, group_size=32, compute_dtype="fp32", scale_dtype="fp32"), device_map="cuda")
    ids = torch.tensor([[5, 9, 33, 2, 71, 9, 9]], device="cuda")
    hf = q._woq_hf_generate(ids, max_new_tokens=12, do_sample=False, repetition_penalty=1.3, pad_token_id=0)
    eng_out = q.generate(ids, max_new_tokens=12, do_sample=False, repetition_penalty=1.3)
    assert hasattr(q, "woq_engine") and torch.equal(eng_out, hf)
    top1 = q.generate(ids, max_new_tokens=12, do_sample=True, top_k=1, temperature=0.7, repetition_penalty=1.3)
    assert torch.equal(top1, hf)
    torch.manual_seed(11)
    a = q.generate(ids, max_new_tokens=24, do_sample=True, temperature=0.9, top_k=8, top_p=0.95, repetition_penalty=1.1)
    torch.manual_seed(11)
    b = q.generate(ids, max_new_tokens=24, do_sample=True, temperature=0.9, top_k=8, top_p=0.95, repetition_penalty=1.1)
    assert torch.equal(a, b) and a.shape == (1, ids.shape[1] + 24)
    # every sampled token is one of the 8 best of HF's penalised scores for its prefix (module-path logits)
    from transformers import RepetitionPenaltyLogitsProcessor

    for i in range(ids.shape[1], a.shape[1]):
        lg = q(a[:, :i]).logits[0, -1].float()
        sc = RepetitionPenaltyLogitsProcessor(1.1)(a[:, :i], lg[None].clone())[0]
        assert int(a[0, i]) in torch.topk(sc, 10).indices.tolist(), i  # 8 + slack for near-ties between the two paths
    first = int(hf[0, ids.shape[1]])
    short = q.generate(ids, max_new_tokens=12, do_sample=False, repetition_penalty=1.3, eos_token_id=first)
    assert short.shape[1] == ids.shape[1] + 1

    class Collect:
        def __init__(self):
            self.items, self.ended = [], False

        def put(self, v):
            self.items.append(v.reshape(-1).tolist())

        def end(self):
            self.ended = True

    st = Collect()
    q.generate(ids, max_new_tokens=5, do_sample=False, repetition_penalty=1.3, streamer=st)
    assert st.ended and st.items[0] == ids[0].tolist() and sum(st.items[1:], []) == hf[0, 7:12].tolist()
    assert q.woq_engine.status() == 0


def test_plain_c_client_device_leg(tmp_path):
    """examples/c_client.c with a GPU: a plain-C99 program that owns its device memory (four HIP runtime entry points
    declared by hand) and goes quantize_to_packed_weight -> read_header -> dequantize_packed_weight -> woq_linear
    through the C ABI alone — no Python, no torch — and checks the reference's own criterion (op == activation x
    dequantised weight) and the RTN step."""
    import os
    import shutil
    import subprocess

    from intel_extension_for_transformers_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "c_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_client.c"), "-L" + libdir, "-lwoq_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    res = subprocess.run([exe, "gpu"], capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert "gpu checks ok" in res.stdout


def test_compute_dtype_int8_is_accepted_and_says_what_it_computes(tmp_path, caplog):
    """compute_dtype="int8" names the reference's dynamic u8-activation cores
    (bestla_weightonly_dispatcher.cpp:131-149). The MI355X path accepts the string and follows the fp32 dequantise ->
    matmul definition at a higher precision — and says so at conversion time (loudly, not silently)."""
    import logging

    from intel_extension_for_transformers_amd.transformers import AutoModelForCausalLM, RtnConfig

    fp = _tiny_llama()
    src = tmp_path / "fp"
    fp.save_pretrained(str(src))
    with caplog.at_level(logging.WARNING):
        qmodel = AutoModelForCausalLM.from_pretrained(str(src), quantization_config=RtnConfig(
            bits=4, group_size=32, compute_dtype="int8"), use_neural_speed=False)
    assert any("compute_dtype='int8'" in r.getMessage() and "HIGHER precision" in r.getMessage() for r in caplog.records)
    ids = torch.tensor([[5, 17, 200, 3]], device="cuda")
    with torch.no_grad():
        logits = qmodel(ids).logits.float()
        ref = _dequantised_twin(qmodel, fp)(ids).logits.float()
    assert (logits - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-5
