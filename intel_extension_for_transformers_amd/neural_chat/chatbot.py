"""build_chatbot / optimize_model (reference neural_chat/chatbot.py:103-321, 384-420) for the HF + weight-only path."""
from .config import PipelineConfig


def build_chatbot(config: PipelineConfig = None):
    """Build the chatbot adapter for `config` and load its model (quantised at load when `optimization_config` is
    a weight-only config). Family dispatch on the model name as in the reference (:141-187)."""
    from .models.base_model import BaseModel, LlamaModel, MistralModel, NeuralChatModel

    if not config:
        config = PipelineConfig()
    name = config.model_name_or_path.lower()
    if "llama" in name:
        adapter = LlamaModel(config.model_name_or_path, config.task)
    elif "neural-chat" in name:
        adapter = NeuralChatModel(config.model_name_or_path, config.task)
    elif "mistral" in name:
        adapter = MistralModel(config.model_name_or_path, config.task)
    else:
        adapter = BaseModel(config.model_name_or_path, config.task)
    parameters = {
        "model_name": config.model_name_or_path,
        "tokenizer_name": config.tokenizer_name_or_path or config.model_name_or_path,
        "device": config.device,
        "use_cache": config.loading_config.use_cache,
        "peft_path": config.loading_config.peft_path,
        "use_neural_speed": config.loading_config.use_neural_speed,
        "gguf_model_path": config.loading_config.gguf_model_path,
        "optimization_config": config.optimization_config,
        "hf_access_token": config.hf_access_token,
        "assistant_model": config.assistant_model,
        "use_vllm": False,
        "vllm_engine_params": None,
    }
    adapter.load_model(parameters)
    return adapter


def optimize_model(model, config, use_neural_speed=False):
    """Reference :384-420 -> Optimization.optimize (pipeline/optimization.py): quantise an already loaded HF model
    per a weight-only config; MixedPrecisionConfig just casts."""
    import torch

    from ..transformers import MixedPrecisionConfig
    from ..transformers.llm.quantization.utils import convert_to_quantized_model

    if use_neural_speed:
        raise NotImplementedError("QBits: Neural Speed is outside the MI355X path")
    if isinstance(config, MixedPrecisionConfig):
        return model.to(getattr(torch, config.dtype))
    config.post_init_hip()
    qmodel = convert_to_quantized_model(model, config, device="cuda")
    qmodel.quantization_config = config
    return qmodel
