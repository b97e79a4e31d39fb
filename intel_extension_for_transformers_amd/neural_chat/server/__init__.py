"""The chat server's HTTP seam (reference: neural_chat/server/neuralchat_server.py:255-330 builds the chatbot from a
YAML file, hands it to every enabled router and starts uvicorn). Here: `create_app(chatbot)` for embedding, and
`python -m intel_extension_for_transformers_amd.neural_chat.server --model <dir> [--bits 4 --group-size 128]` or
`--config neuralchat.yaml` (the reference's server YAML, `pipeline_config_from_yaml`) to serve one weight-only-
quantised model on one MI355X. Other routers (retrieval, voice, image, fine-tuning), plugins, the CLI and the UI are
out of scope."""


def create_app(chatbot):
    from fastapi import FastAPI

    from .restful.textchat_api import router

    router.set_chatbot(chatbot)
    app = FastAPI(title="NeuralChat text-chat on MI355X")
    app.include_router(router)
    return app


def pipeline_config_from_yaml(source):
    """The server YAML of the reference (neuralchat_server.py:150-300, e.g. server/config/neuralchat.yaml) ->
    (PipelineConfig, host, port) for the keys this path reads:

        host / port / model_name_or_path / tokenizer_name_or_path / device
        optimization:
          optimization_type: weight_only | mix_precision
          compute_dtype, weight_dtype           # -> RtnConfig (:277-278)
          use_gptq | use_awq | use_autoround    # -> GPTQConfig / AwqConfig / AutoRoundConfig(bits=4) (:270-275)
          mix_precision_dtype                   # -> MixedPrecisionConfig (:279-280)
          bits, group_size, scale_dtype, scheme # MI355X additions, optional

    `source` is a path or an already parsed dict. Switches of other back ends (ipex_int8, use_neural_speed, use_ggml,
    bits_and_bytes, DeepSpeed / TPP / HPU serving, plugins) raise instead of being ignored."""
    from .. import PipelineConfig
    from ...transformers import AutoRoundConfig, AwqConfig, GPTQConfig, MixedPrecisionConfig, RtnConfig

    if isinstance(source, dict):
        cfg = source
    else:
        import yaml

        with open(source) as f:
            cfg = yaml.safe_load(f) or {}
    for key in ("use_deepspeed", "use_tpp", "habana", "use_hpu_graphs", "use_llm_runtime"):
        if cfg.get(key):
            raise ValueError("QBits: `%s` belongs to another back end; the MI355X server has no such mode" % key)
    for name, block in cfg.items():
        if isinstance(block, dict) and block.get("enable") and name != "optimization":
            raise ValueError("QBits: plugin `%s` is outside the MI355X path (text chat only)" % name)
    tasks = cfg.get("tasks_list")
    if tasks and list(tasks) != ["textchat"]:
        raise ValueError("QBits: only the `textchat` task is served, got %s" % list(tasks))
    opt = cfg.get("optimization") or {}
    for key in ("ipex_int8", "use_neural_speed", "use_ggml", "use_cached_bin", "load_in_4bit"):
        if opt.get(key):
            raise ValueError("QBits: optimization.%s selects another back end; not available on MI355X" % key)
    kind = opt.get("optimization_type") or None
    if kind == "weight_only":
        extra = {k: opt[k] for k in ("group_size", "scale_dtype", "scheme") if opt.get(k) is not None}
        bits = int(opt.get("bits") or 4)
        if opt.get("use_gptq"):
            oc = GPTQConfig(bits=bits, **extra)
        elif opt.get("use_awq"):
            oc = AwqConfig(bits=bits, **extra)
        elif opt.get("use_autoround"):
            oc = AutoRoundConfig(bits=bits, **extra)
        else:
            oc = RtnConfig(bits=bits, compute_dtype=opt.get("compute_dtype") or None,
                           weight_dtype=opt.get("weight_dtype") or None, **extra)
    elif kind == "mix_precision":
        oc = MixedPrecisionConfig(dtype=opt.get("mix_precision_dtype") or "float16")
    elif kind in (None, ""):
        oc = None
    else:
        raise ValueError("QBits: optimization_type `%s` is not available on MI355X (weight_only | mix_precision)" % kind)
    device = cfg.get("device") or "cuda"
    pc = PipelineConfig(model_name_or_path=cfg.get("model_name_or_path") or "Intel/neural-chat-7b-v3-1",
                        tokenizer_name_or_path=cfg.get("tokenizer_name_or_path") or None,
                        device="cuda" if device == "auto" else device, optimization_config=oc, task="chat")
    return pc, cfg.get("host") or "127.0.0.1", int(cfg.get("port") or 8000)


def main(argv=None):
    import argparse

    import uvicorn

    from .. import PipelineConfig, build_chatbot
    from ...transformers import RtnConfig

    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--config", help="server YAML in the reference's format (see pipeline_config_from_yaml)")
    ap.add_argument("--model")
    ap.add_argument("--host", default=None)
    ap.add_argument("--port", type=int, default=None)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--group-size", type=int, default=128)
    ap.add_argument("--scale-dtype", default="fp16")
    a = ap.parse_args(argv)
    if a.config:
        pc, host, port = pipeline_config_from_yaml(a.config)
    elif a.model:
        pc = PipelineConfig(model_name_or_path=a.model, device="cuda",
                            optimization_config=RtnConfig(bits=a.bits, group_size=a.group_size,
                                                          scale_dtype=a.scale_dtype))
        host, port = "127.0.0.1", 8000
    else:
        ap.error("give --config <yaml> or --model <dir>")
    uvicorn.run(create_app(build_chatbot(pc)), host=a.host or host, port=a.port or port)
