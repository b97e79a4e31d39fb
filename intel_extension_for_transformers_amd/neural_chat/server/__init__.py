"""The chat server's HTTP seam (reference: neural_chat/server/neuralchat_server.py:255-330 builds the chatbot from a
YAML file, hands it to every enabled router and starts uvicorn). Here: `create_app(chatbot)` for embedding, and
`python -m intel_extension_for_transformers_amd.neural_chat.server --model <dir> [--bits 4 --group-size 128]` to
serve one weight-only-quantised model on one MI355X. Other routers (retrieval, voice, image, fine-tuning), the YAML
layer, the CLI and the UI are out of scope."""


def create_app(chatbot):
    from fastapi import FastAPI

    from .restful.textchat_api import router

    router.set_chatbot(chatbot)
    app = FastAPI(title="NeuralChat text-chat on MI355X")
    app.include_router(router)
    return app


def main(argv=None):
    import argparse

    import uvicorn

    from .. import PipelineConfig, build_chatbot
    from ...transformers import RtnConfig

    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--model", required=True)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--group-size", type=int, default=128)
    ap.add_argument("--scale-dtype", default="fp16")
    a = ap.parse_args(argv)
    bot = build_chatbot(PipelineConfig(model_name_or_path=a.model, device="cuda",
                                       optimization_config=RtnConfig(bits=a.bits, group_size=a.group_size,
                                                                     scale_dtype=a.scale_dtype)))
    uvicorn.run(create_app(bot), host=a.host, port=a.port)
