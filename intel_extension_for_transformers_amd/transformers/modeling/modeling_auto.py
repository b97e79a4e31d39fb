"""`AutoModelForCausalLM` with the reference's entry points, MI355X backend.

Counterpart of intel_extension_for_transformers/transformers/modeling/modeling_auto.py:
  from_pretrained   :363-905   load an HF model, quantise its linears per `quantization_config`, return the HF model
                               with `.quantization_config` set and `.save_pretrained` patched (:896-903)
  save_low_bit      :209-320   write the quantised model in the INC / optimum tensor format + quantize_config.json
  load_low_bit      :1311-1990 rebuild from that directory
What is different by design: the quantised linears run on the GPU (`device_map="cuda"` is the default here, "cpu"
is rejected — no CPU fallback); `use_neural_speed` is accepted and ignored with a warning (Neural Speed is an x86
runtime); the loader reads safetensors directly instead of HF's private `_load_pretrained_model`
(SURVEY.md §7 hard part iv).
"""
import json
import logging
import os
import types

import torch
import transformers

from ..llm.quantization.nn.modules import QuantizedLinearQBits
from ..llm.quantization.utils import convert_to_quantized_model, pack_weight, replace_linear, unpack_weight
from ..utils.config import (AutoRoundConfig, AwqConfig, GPTQConfig, ITREXQuantizationConfigMixin, RtnConfig,
                            TeqConfig)

logger = logging.getLogger(__name__)

QUANT_CONFIG = "quantize_config.json"  # reference utils/utility.py:34
WEIGHTS_NAME = "model.safetensors"          # what HF save_pretrained writes (the reference saves through it)
LEGACY_WEIGHTS_NAME = "model_woq.safetensors"  # round-1 builds of this package
_BY_METHOD = {"rtn": RtnConfig, "awq": AwqConfig, "teq": TeqConfig, "gptq": GPTQConfig, "autoround": AutoRoundConfig}


def _config_from_dict(d):
    method = d.get("quant_method", "rtn")
    method = getattr(method, "value", method)
    return _BY_METHOD.get(method, RtnConfig).from_dict(d)


def _parse_size(v):
    if isinstance(v, (int, float)):
        return int(v)
    t = str(v).strip().upper()
    for suf, mul in (("GIB", 2 ** 30), ("MIB", 2 ** 20), ("KIB", 2 ** 10), ("GB", 10 ** 9), ("MB", 10 ** 6), ("KB", 10 ** 3)):
        if t.endswith(suf):
            return int(float(t[:-len(suf)]) * mul)
    return int(t)


def save_low_bit(self, save_directory, push_to_hub=False, safe_serialization=True, max_shard_size="5GB", **kwargs):
    """reference modeling_auto.py:209-320 (`convert_model_to_public` :190-205 + `recover_export_model` :99-157, then
    HF `save_pretrained`): every QuantizedLinearQBits is exported as `<name>.qweight / .scales / .qzeros / .g_idx /
    .bias` in the optimum format that `unpack_weight` (utils.py:82-125) reads back; all other tensors are saved as
    they are. Files follow HF's layout, which is what the reference's directories look like: `model.safetensors`, or
    `model-0000i-of-0000n.safetensors` + `model.safetensors.index.json` above `max_shard_size`;
    `safe_serialization=False` writes `pytorch_model.bin`. Plus config.json, quantize_config.json and
    all_checkpoint_keys.json (:289-292)."""
    if push_to_hub:
        raise RuntimeError("push_to_hub is not supported (no network)")
    if kwargs:
        logger.warning("save_low_bit: ignoring %s", sorted(kwargs))
    from safetensors.torch import save_file

    os.makedirs(save_directory, exist_ok=True)
    tensors = {}
    quantized = set()
    for name, mod in self.named_modules():
        if isinstance(mod, QuantizedLinearQBits):
            int_w, scales, zeros, g_idx = mod.recover_qparms_kn()
            if mod.bits == 8:  # on disk 8-bit values live in the unsigned domain, q + 128 (utils.py:103-124 undoes it)
                int_w = int_w.to(torch.int16) + 128
                zeros = None if zeros is None else zeros.to(torch.int16) + 128
            qweight, sc16, qzeros = pack_weight(int_w, scales, zeros, bits=mod.bits)
            tensors[name + ".qweight"] = qweight.cpu().contiguous()
            tensors[name + ".scales"] = (scales if mod.scale_dtype == "fp32" else sc16).cpu().contiguous()
            if qzeros is not None:
                tensors[name + ".qzeros"] = qzeros.cpu().contiguous()
            if g_idx is not None:
                tensors[name + ".g_idx"] = g_idx.cpu().contiguous()
            if mod.bias is not None:
                tensors[name + ".bias"] = mod.bias.detach().cpu().contiguous()
            quantized.add(name)
    seen_ptr = {}
    for key, t in self.state_dict().items():
        owner = key.rsplit(".", 1)[0]
        if owner in quantized:
            continue
        t = t.detach()
        ptr = t.data_ptr()
        if ptr in seen_ptr and t.numel():  # tied weights (embed / lm_head): store once, record the alias
            seen_ptr[ptr].append(key)
            continue
        seen_ptr[ptr] = [key]
        tensors[key] = t.cpu().contiguous()
    aliases = {names[0]: names[1:] for names in seen_ptr.values() if len(names) > 1}
    meta = {"format": "pt", "aliases": json.dumps(aliases), "quantized": json.dumps(sorted(quantized))}
    if not safe_serialization:
        torch.save(tensors, os.path.join(save_directory, "pytorch_model.bin"))
    else:
        limit = _parse_size(max_shard_size)
        shards, cur, cur_bytes = [], {}, 0
        for key in sorted(tensors):
            nbytes = tensors[key].numel() * tensors[key].element_size()
            if cur and cur_bytes + nbytes > limit:
                shards.append(cur)
                cur, cur_bytes = {}, 0
            cur[key] = tensors[key]
            cur_bytes += nbytes
        shards.append(cur)
        if len(shards) == 1:
            save_file(shards[0], os.path.join(save_directory, WEIGHTS_NAME), metadata=meta)
        else:
            weight_map, total = {}, 0
            for i, shard in enumerate(shards):
                fname = "model-%05d-of-%05d.safetensors" % (i + 1, len(shards))
                save_file(shard, os.path.join(save_directory, fname), metadata=meta)
                for key, t in shard.items():
                    weight_map[key] = fname
                    total += t.numel() * t.element_size()
            with open(os.path.join(save_directory, WEIGHTS_NAME + ".index.json"), "w") as f:
                json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
    self.config.save_pretrained(save_directory)
    qcfg = self.quantization_config
    qcfg.save_pretrained(save_directory)
    with open(os.path.join(save_directory, "all_checkpoint_keys.json"), "w") as f:  # modeling_auto.py:289-292
        json.dump({"all_checkpoint_keys": sorted(tensors.keys())}, f)


class _ShellLinear(torch.nn.Module):
    """Holder of optimum-format tensors between file and repack (the role INC's WeightOnlyLinear shells play in
    `build_woq_model`, reference modeling_auto.py:160-187)."""

    def __init__(self, in_features, out_features, tensors, awq_gemm=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.awq_gemm = awq_gemm  # AutoAWQ "GEMM" packing (along N, interleaved nibble order) instead of optimum's
        self.qweight = tensors["qweight"]
        self.scales = tensors["scales"]
        self.qzeros = tensors.get("qzeros")
        self.g_idx = tensors.get("g_idx")
        self.bias = tensors.get("bias")


def _read_safetensors(paths):
    from safetensors import safe_open

    tensors, meta = {}, {}
    for path in paths:
        with safe_open(path, framework="pt", device="cpu") as f:
            meta.update(f.metadata() or {})
            for k in f.keys():
                tensors[k] = f.get_tensor(k)
    return tensors, meta


def _load_packed(d, qcfg, tensors, quantized, aliases, device, awq_gemm=False):
    """config -> empty model -> packed-linear shells -> repack on the device (the tail of reference
    modeling_auto.py:1311-1990, shared by our own saved format and by HF-hub GPTQ checkpoints)."""
    config = transformers.AutoConfig.from_pretrained(d)
    if hasattr(config, "quantization_config"):  # keep HF from looking for its own quantizer back end
        try:
            delattr(config, "quantization_config")
        except AttributeError:
            config.quantization_config = None
    with torch.device("meta"):
        model = transformers.AutoModelForCausalLM.from_config(config)
    for name in quantized:
        parent_name, _, leaf = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        old = getattr(parent, leaf)
        if hasattr(old, "nf"):
            k, n = old.weight.shape[0], old.nf
        else:
            k, n = old.in_features, old.out_features
        parts = {s: tensors[name + "." + s].to(device) for s in ("qweight", "scales", "qzeros", "g_idx", "bias")
                 if name + "." + s in tensors}
        parent._modules[leaf] = _ShellLinear(k, n, parts, awq_gemm=awq_gemm)
    skip = [n for n, m in model.named_modules()
            if isinstance(m, torch.nn.Linear) or type(m).__name__ == "Conv1D"]  # whatever was left unquantised
    replace_linear(model, skip, None, qcfg, device=device)
    state = {k: v for k, v in tensors.items() if not any(k.startswith(q + ".") for q in quantized)}
    for first, rest in aliases.items():
        for r in rest:
            state[r] = state[first]
    missing, unexpected = model.load_state_dict(state, strict=False, assign=True)
    missing = [m for m in missing if not any(m.startswith(q + ".") for q in quantized)]
    if missing:
        logger.warning("packed checkpoint: tensors not found: %s", missing[:8])
    model.tie_weights()
    _materialise_buffers(model, device)
    model.to(device)
    model.eval()
    return _finish(model, qcfg)


def _checkpoint_tensors(d):
    """The weight files of an HF-layout directory, whichever of the standard forms it uses: sharded safetensors
    (index json), one model.safetensors, pytorch_model.bin (sharded or not), or round 1's model_woq.safetensors."""
    for index in (WEIGHTS_NAME + ".index.json", "pytorch_model.bin.index.json"):
        path = os.path.join(d, index)
        if os.path.isfile(path):
            with open(path) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            if index.startswith("pytorch_model"):
                tensors = {}
                for fn in files:
                    tensors.update(torch.load(os.path.join(d, fn), map_location="cpu", weights_only=True))
                return tensors, {}
            return _read_safetensors([os.path.join(d, fn) for fn in files])
    for fn in (WEIGHTS_NAME, LEGACY_WEIGHTS_NAME):
        if os.path.isfile(os.path.join(d, fn)):
            return _read_safetensors([os.path.join(d, fn)])
    if os.path.isfile(os.path.join(d, "pytorch_model.bin")):
        return torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu", weights_only=True), {}
    raise FileNotFoundError("QBits: no model.safetensors / pytorch_model.bin (or their index) in %s" % d)


def load_low_bit(pretrained_model_name_or_path, device="cuda", **kwargs):
    """reference modeling_auto.py:1311-1990: a directory written by `save_low_bit` — this package's or the
    reference's (same HF file layout, same optimum tensor names: the packed linears are the keys ending in .qweight,
    tied weights are re-tied from the config)."""
    d = str(pretrained_model_name_or_path)
    with open(os.path.join(d, QUANT_CONFIG)) as f:
        qcfg = _config_from_dict(json.load(f))
    qcfg.post_init_hip()
    tensors, meta = _checkpoint_tensors(d)
    aliases = json.loads(meta.get("aliases", "{}"))
    quantized = json.loads(meta["quantized"]) if "quantized" in meta else sorted(
        k[:-len(".qweight")] for k in tensors if k.endswith(".qweight"))
    return _load_packed(d, qcfg, tensors, quantized, aliases, device)


def _hf_gptq_dir(path):
    """A Hugging Face GPTQ or AWQ checkpoint directory: config.json carries quantization_config.quant_method ==
    "gptq" (AutoGPTQ / optimum writer: `<linear>.qweight / .qzeros / .scales / .g_idx` in the packing `unpack_weight`
    reads — int32 words of 8 nibbles along K, zeros stored as zp - 1, utils.py:82-125) or "awq" (AutoAWQ "GEMM"
    writer: packed along N, see utils.unpack_awq_gemm)."""
    cfg_file = os.path.join(str(path), "config.json")
    if not os.path.isfile(cfg_file) or os.path.isfile(os.path.join(str(path), QUANT_CONFIG)):
        return None
    with open(cfg_file) as f:
        q = (json.load(f).get("quantization_config") or {})
    return q if str(q.get("quant_method", "")).lower() in ("gptq", "awq") else None


def load_hf_gptq(path, q, device="cuda"):
    """a pre-quantised GPTQ / AWQ checkpoint straight to the GPU layout, act-order
    (desc_act) included — the rows are regrouped at load and the activation shuffle is applied by the kernels."""
    d = str(path)
    awq = str(q.get("quant_method", "")).lower() == "awq"
    if awq:
        if str(q.get("version", "gemm")).lower() != "gemm" or int(q.get("bits", q.get("w_bit", 4))) != 4:
            raise RuntimeError("QBits: only 4-bit AutoAWQ 'GEMM' checkpoints are read")
        qcfg = AwqConfig(bits=4, group_size=int(q.get("group_size", q.get("q_group_size", 128))),
                         zero_point=bool(q.get("zero_point", True)), scale_dtype="fp16")
    else:
        qcfg = GPTQConfig(bits=int(q.get("bits", 4)), group_size=int(q.get("group_size", 128)),
                          sym=bool(q.get("sym", True)), desc_act=bool(q.get("desc_act", False)),
                          static_groups=bool(q.get("static_groups", False)), scale_dtype="fp16")
    qcfg.post_init_hip()
    index = os.path.join(d, "model.safetensors.index.json")
    if os.path.isfile(index):
        with open(index) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = ["model.safetensors"]
    tensors, _ = _read_safetensors([os.path.join(d, f) for f in files])
    quantized = sorted(k[:-len(".qweight")] for k in tensors if k.endswith(".qweight"))
    if not quantized:
        raise RuntimeError("QBits: no packed linears (*.qweight) found in %s" % d)
    return _load_packed(d, qcfg, tensors, quantized, {}, device, awq_gemm=awq)


def _materialise_buffers(model, device):
    """Non-persistent buffers (rotary inv_freq ...) are not in the checkpoint: rebuild the modules that own meta
    buffers from the config."""
    for name, mod in list(model.named_modules()):
        if any(b is not None and b.is_meta for b in mod.buffers(recurse=False)):
            try:
                fresh = type(mod)(config=model.config)
            except Exception:
                fresh = type(mod)(model.config)
            parent_name, _, leaf = name.rpartition(".")
            (model.get_submodule(parent_name) if parent_name else model)._modules[leaf] = fresh.to(device)


def _engine_generate(self, inputs=None, generation_config=None, **kwargs):
    """`model.generate` with the fused native engine underneath when the request is one it covers — a single
    sequence, greedy, one beam, no logits processing beyond the default — and HF's own generate (over the
    QuantizedLinearQBits modules) for everything else. Same return value as HF for the covered case: LongTensor
    [1, prompt + new]. The reference has no counterpart (its CPU path always runs the module-by-module loop); the
    precedent for swapping the loop after from_pretrained is ipex.optimize_transformers (docs/weightonlyquant.md:199)."""
    # reference greedy_search.py:148-150,374-377,408-409: config.token_latency (or model.token_latency) makes generate
    # return (ids, latency_list), one wall-clock entry per generated token, the first one being the prompt pass
    want_latency = bool(getattr(self.config, "token_latency", False) or getattr(self, "token_latency", False))
    raw_hf_generate = self._woq_hf_generate

    def hf_generate(*a, **kw):
        """HF's loop over the quantised modules; with token_latency the whole call is timed and every new token
        gets the average (HF's loop has no per-token hook without the reference's patched greedy_search)."""
        if not want_latency:
            return raw_hf_generate(*a, **kw)
        import time

        tic = time.time()
        res = raw_hf_generate(*a, **kw)
        torch.cuda.synchronize()
        seq = res if torch.is_tensor(res) else res.sequences
        n_prompt = ids.shape[1] if torch.is_tensor(ids) and ids.dim() == 2 else 0
        n_new = max(int(seq.shape[1]) - int(n_prompt), 1)
        return res, [(time.time() - tic) / n_new] * n_new

    ids = inputs if inputs is not None else kwargs.get("input_ids")
    gc = generation_config if generation_config is not None else self.generation_config
    opt = lambda k, d=None: kwargs[k] if k in kwargs else getattr(gc, k, d)  # noqa: E731
    # sampling (temperature / top-k / top-p) and the repetition penalty — the reference's NeuralChat defaults
    # (neural_chat/config.py:400-409) — ride the engine too, the next token chosen on the device (runtime.engine.
    # DeviceSampler: HF's processors / warpers in HF's order); any other logits processing keeps HF's loop
    sampled = bool(opt("do_sample", False)) or (opt("repetition_penalty", 1.0) or 1.0) != 1.0
    plain_sampling = ((opt("typical_p", 1.0) or 1.0) == 1.0 and not opt("epsilon_cutoff", 0.0) and not opt("eta_cutoff", 0.0)
                      and opt("min_p") is None and opt("penalty_alpha") is None
                      and (opt("encoder_repetition_penalty", 1.0) or 1.0) == 1.0 and not opt("renormalize_logits", False)
                      and all(opt(k) is None for k in ("exponential_decay_length_penalty", "suppress_tokens",
                                                       "begin_suppress_tokens", "forced_bos_token_id", "forced_eos_token_id",
                                                       "sequence_bias", "guidance_scale")))
    simple = (torch.is_tensor(ids) and ids.dim() == 2 and ids.shape[0] == 1 and ids.shape[1] >= 1
              and (not sampled or plain_sampling) and (opt("num_beams", 1) or 1) == 1
              and (opt("num_return_sequences", 1) or 1) == 1
              and not opt("no_repeat_ngram_size", 0) and not opt("bad_words_ids") and not opt("force_words_ids")
              and not opt("min_new_tokens", 0) and not opt("min_length", 0)
              and not any(kwargs.get(k) is not None for k in ("logits_processor", "stopping_criteria",
                                                              "prefix_allowed_tokens_fn", "assistant_model",
                                                              "past_key_values", "inputs_embeds"))
              and not kwargs.get("return_dict_in_generate", False) and not kwargs.get("output_scores", False))
    mask = kwargs.get("attention_mask")
    if simple and mask is not None and not bool(torch.all(mask == 1)):
        simple = False
    if not simple or getattr(self, "_woq_engine_off", False):
        return hf_generate(inputs, generation_config=generation_config, **kwargs) if inputs is not None else \
            hf_generate(generation_config=generation_config, **kwargs)
    n_in = int(ids.shape[1])
    max_new = opt("max_new_tokens")
    if max_new is None:
        max_new = max(int(opt("max_length", 20)) - n_in, 0)
    eng = getattr(self, "woq_engine", None)
    if eng is None or eng.cfg.max_ctx < n_in + max_new:
        from ...runtime.engine import optimize_transformers

        want = n_in + max_new
        ctx = int(min(max(want, 512), getattr(self.config, "max_position_embeddings", want) or want))
        if ctx < want:
            return hf_generate(inputs, generation_config=generation_config, **kwargs)
        try:
            eng = optimize_transformers(self, max_ctx=1 << (ctx - 1).bit_length())
        except RuntimeError:  # not a Llama-class int4 model: the module path serves it
            self._woq_engine_off = True
            return hf_generate(inputs, generation_config=generation_config, **kwargs)
    if max_new < 1:
        return (ids, []) if want_latency else ids
    eos = opt("eos_token_id")
    eos = set() if eos is None else set(eos if isinstance(eos, (list, tuple)) else [eos])
    streamer = kwargs.get("streamer")
    prompt = ids[0].tolist()
    if streamer is not None:
        streamer.put(ids.cpu())
    import time

    def run_sampled():
        """prompt pass + steps whose next token the device sampler picks; returns (tokens, per-token latency)."""
        from ...runtime.engine import DeviceSampler, generate_sampled

        sampler = DeviceSampler(do_sample=opt("do_sample", False), temperature=opt("temperature", 1.0),
                                top_k=opt("top_k", 0), top_p=opt("top_p", 1.0),
                                repetition_penalty=opt("repetition_penalty", 1.0))
        latency, mark = [], [time.time()]

        def on_tokens(new):
            now = time.time()
            latency.extend([(now - mark[0]) / len(new)] * len(new))
            mark[0] = now
            if streamer is not None:
                for t in new:
                    streamer.put(torch.tensor([t]))

        out = generate_sampled(eng, prompt, max_new, sampler, eos=eos, on_tokens=on_tokens,
                               burst=1 if streamer is not None else 16)
        return out, latency

    def run_once():
        """prompt pass + chained decode steps; returns (tokens, per-token latency)."""
        if sampled:
            return run_sampled()
        latency = []
        tic = time.time()
        for s0 in range(0, n_in, 2048):
            eng.prefill(prompt[s0:s0 + 2048], start_pos=s0, greedy=True)
        eng.tune_attn_for(n_in + max_new)
        if max_new > 1:
            eng.prepare_decode(greedy=True)  # a captured graph only in "graph" launch mode (engine.py LAUNCH)
        # Tokens are read back in bursts: the steps chain on the device and log their tokens (engine.token_log), so the
        # host synchronises once per burst instead of once per token. A stop token is noticed at the end of its burst —
        # the few steps run past it are discarded. With a streamer the burst is one token (latency first).
        out = [int(eng.token.item())]  # the prompt pass's token (the .item() synchronises)
        latency.append(time.time() - tic)
        if streamer is not None:
            streamer.put(torch.tensor(out))
        burst = 1 if streamer is not None else 16
        log = eng.token_log()
        done = out[0] in eos
        while not done and len(out) < max_new:
            k = min(burst, max_new - len(out))
            p0 = n_in + len(out) - 1  # position the next step feeds
            tic = time.time()
            eng.replay(k)
            new = log[p0:p0 + k].tolist()  # one host synchronisation per burst
            per_token = (time.time() - tic) / k  # the burst's steps chain on the device: its tokens share the average
            for t in new:
                latency.append(per_token)
                out.append(t)
                if streamer is not None:
                    streamer.put(torch.tensor([t]))
                if t in eos:
                    done = True
                    break
        return out, latency

    out, latency = run_once()
    # The decode step reports what it cannot raise from the device through a sticky status word; the host has just
    # synchronised on the last burst, so one more 4-byte read is free. bit 0: an attention workgroup of the fused
    # qkv + attention launch gave up waiting for its head's q / k / v (GPU preempted / shared / under a debugger for
    # longer than the hand-off's 20 ms bound) and went on with stale values — the tokens since then and the KV rows of
    # those positions are wrong. bit 1: a step ran at or beyond max_ctx. Neither may come back as a normal result.
    st = eng.status()
    if st != 0:
        eng.clear_status()
        if st & 1:
            eng.set_fuse_attn(False)  # the two-launch form has no in-launch wait; the prompt pass rewrites the cache
            logger.warning("QBits: the fused qkv + attention launch timed out waiting for a hand-off (engine status %d); "
                           "the engine falls back to separate launches", st)
        if st < 0 or (st & ~1) or streamer is not None:
            if streamer is not None:
                streamer.end()
            raise RuntimeError("QBits: the decode engine reported status %d (bit 0: in-launch hand-off timed out, "
                               "bit 1: position beyond max_ctx)%s" % (st, "; tokens already streamed are not reliable"
                                                                     if streamer is not None else ""))
        out, latency = run_once()  # same request, separate launches
        st = eng.status()
        if st != 0:
            eng.clear_status()
            raise RuntimeError("QBits: the decode engine reported status %d on the retry" % st)
    if streamer is not None:
        streamer.end()
    result = torch.cat([ids, torch.tensor([out], dtype=ids.dtype, device=ids.device)], dim=1)
    return (result, latency[:len(out)]) if want_latency else result


def _finish(model, qcfg):
    """reference modeling_auto.py:896-903, plus the engine-backed `generate` for Llama-class int4 models."""
    qcfg.remove_redundant_parameters()
    model.quantization_config = qcfg
    model.config.quantization_config = qcfg.to_dict()
    model.save_pretrained = types.MethodType(save_low_bit, model)
    if getattr(model.config, "model_type", "") in ("llama", "mistral") and getattr(qcfg, "bits", 4) == 4 \
            and hasattr(model, "generate") and not hasattr(model, "_woq_hf_generate"):
        model._woq_hf_generate = model.generate
        model.generate = types.MethodType(_engine_generate, model)
    return model


class _BaseAutoModelClass:
    ORIG_MODEL = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """`from_pretrained(name_or_dir | nn.Module, quantization_config=RtnConfig(...) | load_in_4bit=True,
        device_map="cuda", use_neural_speed=False, ...)` — reference modeling_auto.py:363-905."""
        device_map = kwargs.pop("device_map", "cuda")
        device = "cuda" if device_map in ("auto", "cuda", None) else str(device_map)
        if device == "cpu":
            raise RuntimeError("QBits: the MI355X backend has no CPU path; use device_map='cuda'")
        if kwargs.pop("use_neural_speed", False) or kwargs.pop("use_llm_runtime", False):
            logger.warning("use_neural_speed is an x86 runtime switch; ignored on the MI355X backend")
        if kwargs.pop("use_vllm", None):
            # reference modeling_auto.py:364-480 builds a vllm.LLM, swaps its parallel linears for torch linears,
            # reloads the weights and quantises them. vLLM is not part of this image; what this build keeps of that
            # seam is the module-side contract (QuantizedLinearQBits.forward returns (output, None) under
            # backend=use_vllm), so the same surgery works where vLLM exists.
            raise RuntimeError("QBits: use_vllm needs the vllm package, which this build does not ship; the "
                               "quantised linear honours vLLM's (output, bias) return convention under "
                               "backend=use_vllm")
        qcfg = kwargs.pop("quantization_config", None)
        load_in_4bit = kwargs.pop("load_in_4bit", False)
        load_in_8bit = kwargs.pop("load_in_8bit", False)
        kwargs.pop("use_cpu", None)
        kwargs.pop("use_xpu", None)
        if isinstance(pretrained_model_name_or_path, (str, os.PathLike)) and os.path.isfile(
                os.path.join(str(pretrained_model_name_or_path), QUANT_CONFIG)):
            return load_low_bit(pretrained_model_name_or_path, device=device)  # :598-657
        if isinstance(pretrained_model_name_or_path, (str, os.PathLike)) and os.path.isdir(
                str(pretrained_model_name_or_path)):
            gq = _hf_gptq_dir(pretrained_model_name_or_path)
            if gq is not None:
                return load_hf_gptq(pretrained_model_name_or_path, gq, device=device)
        if qcfg is None and (load_in_4bit or load_in_8bit):  # :717-741: load_in_{4,8}bit -> RtnConfig(bits=4 | 8)
            qcfg = RtnConfig(bits=4 if load_in_4bit else 8, compute_dtype=kwargs.pop("compute_dtype", None),
                             weight_dtype=kwargs.pop("weight_dtype", None), scale_dtype=kwargs.pop("scale_dtype", None))
        if isinstance(pretrained_model_name_or_path, torch.nn.Module):
            model = pretrained_model_name_or_path
        else:
            kwargs.setdefault("torch_dtype", torch.float32)  # quantise from fp32 (utils.py:539-544)
            kwargs["low_cpu_mem_usage"] = True
            model = cls.ORIG_MODEL.from_pretrained(pretrained_model_name_or_path, *model_args, **kwargs)
        if qcfg is None:
            return model.to(device)
        if not isinstance(qcfg, ITREXQuantizationConfigMixin):
            raise ValueError("quantization_config must be one of RtnConfig / AwqConfig / TeqConfig / GPTQConfig / "
                             "AutoRoundConfig")
        qcfg.post_init_hip()
        model = convert_to_quantized_model(model, qcfg, device=device)
        return _finish(model, qcfg)

    @classmethod
    def load_low_bit(cls, pretrained_model_name_or_path, *args, **kwargs):
        return load_low_bit(pretrained_model_name_or_path, **kwargs)


class AutoModelForCausalLM(_BaseAutoModelClass):
    ORIG_MODEL = transformers.AutoModelForCausalLM


class AutoModel(_BaseAutoModelClass):
    ORIG_MODEL = transformers.AutoModel


class AutoModelForSeq2SeqLM(_BaseAutoModelClass):
    ORIG_MODEL = transformers.AutoModelForSeq2SeqLM
