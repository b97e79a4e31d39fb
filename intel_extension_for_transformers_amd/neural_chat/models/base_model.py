"""Model adapters behind build_chatbot (reference: neural_chat/models/base_model.py:64-420 and model_utils.py
load_model :578-1000 / predict_stream :1061-1380 / predict :1383-1600, reduced to the HF + weight-only-quantised path).

Generation runs on the GPU in one of two ways:
 * greedy, single beam, Llama-class model  -> the fused decode engine (runtime/engine.py: prompt pass on the MFMA
   GEMMs, then hipGraph-replayed decode steps), tokens streamed as they are produced;
 * anything else (sampling, beams, other architectures) -> `model.generate` of the HF model whose linears are
   QuantizedLinearQBits modules, with a TextIteratorStreamer on a worker thread exactly like the reference
   (model_utils.py:1249).
"""
import time
from threading import Thread

from ..config import GenerationConfig
from ..prompts import get_conv_template


class BaseModel:
    def __init__(self, model_name="", task=""):
        self.model_name = model_name
        self.task = task
        self.model = None
        self.tokenizer = None
        self.engine = None
        self.device = "cuda"
        self.conv_template = None
        self.use_cache = True

    # ---- reference BaseModel surface -------------------------------------------------------------------------
    def match(self):
        return True

    def get_default_conv_template(self):
        return get_conv_template("raw")

    def load_model(self, kwargs: dict):
        """kwargs keys as the reference builds them in chatbot.py (model_name, tokenizer_name, device,
        optimization_config, use_cache, ...)."""
        import torch
        from transformers import AutoTokenizer

        from ...transformers import AutoModelForCausalLM, MixedPrecisionConfig

        self.model_name = kwargs["model_name"]
        self.device = kwargs.get("device", "cuda")
        self.use_cache = kwargs.get("use_cache", True)
        if kwargs.get("use_neural_speed") or kwargs.get("use_vllm") or kwargs.get("gguf_model_path"):
            raise NotImplementedError("QBits: Neural Speed / GGUF / vLLM back ends are outside the MI355X path")
        if self.device != "cuda":
            raise RuntimeError("QBits: the MI355X backend has no %s path; use device='cuda'" % self.device)
        opt = kwargs.get("optimization_config")
        self.tokenizer = AutoTokenizer.from_pretrained(kwargs.get("tokenizer_name") or self.model_name)
        if opt is None or isinstance(opt, MixedPrecisionConfig):
            dt = getattr(torch, (opt.dtype if opt is not None else "float16"))
            self.model = AutoModelForCausalLM.from_pretrained(self.model_name, torch_dtype=dt, device_map="cuda")
        else:  # weight-only quantisation at load time, reference model_utils.py:796-822
            self.model = AutoModelForCausalLM.from_pretrained(self.model_name, quantization_config=opt,
                                                              device_map="cuda")
        self.model.eval()
        self.conv_template = self.get_default_conv_template()
        self._try_engine()

    def _try_engine(self):
        """Build the fused decode engine when the model is a quantised Llama-class decoder; otherwise stay on the
        module path (not an error)."""
        self.engine = None
        if not hasattr(self.model, "quantization_config"):
            return
        try:
            from ...runtime.engine import optimize_transformers

            ctx = int(min(getattr(self.model.config, "max_position_embeddings", 2048) or 2048, 8192))
            self.engine = optimize_transformers(self.model, max_ctx=ctx)
        except RuntimeError:
            self.engine = None

    def prepare_prompt(self, query, config=None):
        conv = self.conv_template.copy()
        if conv.roles[0] and conv.roles[0] in query and conv.roles[1] in query:
            return query  # the caller already applied a template (reference base_model.py:174-180)
        conv.append_message(conv.roles[0], query)
        conv.append_message(conv.roles[1], None)
        return conv.get_prompt()

    def predict_stream(self, query, origin_query="", config=None):
        """-> (generator of text pieces, link) like the reference (base_model.py:150-273: `return response, link`;
        `link` is the retrieval plugin's source list, always empty here — plugins are out of scope). With
        config.return_stats the reference's trailing stats block follows the text (model_utils.py:1340-1380,
        format v1 / v2)."""
        config = config or GenerationConfig()
        prompt = self.prepare_prompt(query, config)
        return self._stream(prompt, config), []

    def predict(self, query, origin_query="", config=None):
        config = config or GenerationConfig()
        prompt = self.prepare_prompt(query, config)
        stats = config.return_stats
        config.return_stats = False
        try:
            return "".join(self._stream(prompt, config))
        finally:
            config.return_stats = stats

    # ---- generation ----------------------------------------------------------------------------------------------
    def _stream(self, prompt, config):
        ids = self.tokenizer(prompt, return_tensors="pt").input_ids
        n_in = int(ids.shape[1])
        t_start = time.time()
        first = [None]
        n_out = [0]
        # one sequence, one beam: the fused engine — greedy chains on the device; sampling (the reference's default:
        # do_sample, temperature, top_k, top_p) and the repetition penalty pick the next token with the device sampler
        engine_ok = config.num_beams == 1 and (config.num_return_sequences or 1) == 1 and not config.bad_words_ids \
            and not config.force_words_ids

        def pieces():
            if engine_ok and self.engine is not None and n_in + config.max_new_tokens <= self.engine.cfg.max_ctx:
                yield from self._engine_stream(ids[0].tolist(), config, n_out)
            else:
                yield from self._hf_stream(ids, config, n_out)

        for text in pieces():
            if not text:
                continue
            if first[0] is None:
                first[0] = time.time()
            yield text
        if config.return_stats:
            dur = int((time.time() - t_start) * 1000)
            ftl = int(((first[0] or time.time()) - t_start) * 1000)
            per = dur / n_out[0] if n_out[0] else 0
            if config.format_version == "v1":
                yield "END_OF_STREAM_STATS={}".format({"input_token_len": n_in, "output_token_len": n_in + n_out[0],
                                                       "duration": dur, "first_token_latency": ftl,
                                                       "msecond_per_token": per})
            else:
                stats = {"input_token_len": str(n_in), "output_token_len": str(n_in + n_out[0]),
                         "duration": "%d ms" % dur, "first_token_latency": "%d ms" % ftl,
                         "msecond_per_token": "%s ms" % per}
                yield "\n| {:<22} | {:<27} |\n".format("Key", "Value")
                yield "| " + "-" * 22 + " | " + "-" * 27 + " |" + "\n"
                for k, v in stats.items():
                    yield "| {:<22} | {:<27} |\n".format(k, v)

    def _engine_stream(self, ids, config, n_out):
        eng, tok = self.engine, self.tokenizer
        eos = tok.eos_token_id
        if config.do_sample or (config.repetition_penalty or 1.0) != 1.0:
            yield from self._engine_stream_sampled(ids, config, n_out)
            return
        for s0 in range(0, len(ids), 2048):
            eng.prefill(ids[s0:s0 + 2048], start_pos=s0, greedy=True)
        eng.tune_attn_for(len(ids) + config.max_new_tokens)
        eng.prepare_decode(greedy=True)  # a captured graph only in "graph" launch mode (engine.py LAUNCH)
        out, shown = [], ""
        for i in range(config.max_new_tokens):
            t = int(eng.token.item())
            if eos is not None and t == eos:
                break
            out.append(t)
            n_out[0] += 1
            text = tok.decode(out, skip_special_tokens=True)
            if not text.endswith("�"):  # hold back incomplete multi-byte pieces, like TextIteratorStreamer
                yield text[len(shown):]
                shown = text
            if i + 1 < config.max_new_tokens:
                eng.replay(1)

    def _engine_stream_sampled(self, ids, config, n_out):
        """The reference's default request (sampling + repetition penalty) on the fused engine: tokens are chosen on the
        device (runtime.engine.DeviceSampler) and handed to the text stream one at a time."""
        from ...runtime.engine import DeviceSampler, iter_sampled

        eng, tok = self.engine, self.tokenizer
        eos = tok.eos_token_id
        sampler = DeviceSampler(do_sample=config.do_sample, temperature=config.temperature, top_k=config.top_k,
                                top_p=config.top_p, repetition_penalty=config.repetition_penalty)
        out, shown = [], ""
        for new in iter_sampled(eng, ids, config.max_new_tokens, sampler, eos=() if eos is None else (eos,), burst=1):
            for t in new:
                if eos is not None and t == eos:
                    continue
                out.append(t)
                n_out[0] += 1
            text = tok.decode(out, skip_special_tokens=True)
            if not text.endswith("\ufffd"):  # hold back incomplete multi-byte pieces, like TextIteratorStreamer
                yield text[len(shown):]
                shown = text

    def _hf_stream(self, ids, config, n_out):
        import torch
        from transformers import TextIteratorStreamer

        streamer = TextIteratorStreamer(self.tokenizer, skip_prompt=True, skip_special_tokens=True)
        gen = dict(max_new_tokens=config.max_new_tokens, do_sample=config.do_sample, num_beams=config.num_beams,
                   use_cache=config.use_cache, repetition_penalty=config.repetition_penalty,
                   num_return_sequences=config.num_return_sequences, pad_token_id=self.tokenizer.eos_token_id)
        if config.do_sample:
            gen.update(temperature=config.temperature, top_k=config.top_k, top_p=config.top_p)
        err = []
        result = []

        def work():
            try:
                with torch.no_grad():
                    result.append(self.model.generate(ids.to("cuda"), streamer=streamer, **gen))
            except Exception as e:  # surfaced on the consumer side, like the reference's errors_queue
                err.append(e)
                streamer.end()

        th = Thread(target=work)
        th.start()
        for text in streamer:
            yield text
        th.join()
        if err:
            raise err[0]
        if result:
            n_out[0] = int(result[0].shape[-1]) - int(ids.shape[1])


class LlamaModel(BaseModel):
    def match(self):
        return "llama" in self.model_name.lower()

    def get_default_conv_template(self):
        return get_conv_template("llama-2")


class MistralModel(BaseModel):
    def match(self):
        return "mistral" in self.model_name.lower()

    def get_default_conv_template(self):
        return get_conv_template("mistral")


class NeuralChatModel(BaseModel):
    def match(self):
        return "neural-chat" in self.model_name.lower()

    def get_default_conv_template(self):
        return get_conv_template("neural-chat-7b-v3")
