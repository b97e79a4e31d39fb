"""`intel_extension_for_transformers.neural_chat` serving seam on the MI355X int4 path.

Only the path north_star names: `PipelineConfig(optimization_config=RtnConfig(...))` -> `build_chatbot` ->
`predict` / `predict_stream` (reference: neural_chat/chatbot.py:103-321, models/base_model.py:101-420,
models/model_utils.py:1061-1380). Plugins, the REST/OpenAI servers, finetuning, audio/image tasks and the
non-HF back ends (OpenAI, vLLM, Neural Speed/GGUF, HPU) are out of scope.
"""
from .chatbot import build_chatbot, optimize_model  # noqa: F401
from .config import GenerationConfig, LoadingModelConfig, PipelineConfig  # noqa: F401

__all__ = ["build_chatbot", "optimize_model", "PipelineConfig", "GenerationConfig", "LoadingModelConfig"]
