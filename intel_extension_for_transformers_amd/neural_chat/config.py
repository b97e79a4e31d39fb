"""Configuration objects of the chat pipeline — field names and defaults of the reference's neural_chat/config.py
(GenerationConfig :400-423, LoadingModelConfig :426-436, PipelineConfig :466-517) for the fields this path reads."""
from dataclasses import dataclass
from typing import List


@dataclass
class GenerationConfig:
    device: str = "cuda"
    temperature: float = 0.1
    top_k: int = 40
    top_p: float = 0.75
    repetition_penalty: float = 1.1
    num_beams: int = 1
    max_new_tokens: int = 256
    do_sample: bool = True
    num_return_sequences: int = 1
    bad_words_ids: List[int] = None
    force_words_ids: List[int] = None
    use_cache: bool = True
    return_stats: bool = False
    format_version: str = "v2"
    task: str = ""


@dataclass
class LoadingModelConfig:
    peft_path: str = None
    use_cache: bool = True
    world_size: int = 1
    use_neural_speed: bool = False  # accepted for signature parity; must stay False here
    gguf_model_path: str = None


def get_device_type():
    """Reference utils/common.py get_device_type picks cuda > xpu > hpu > cpu; here there is one backend."""
    import torch

    if torch.cuda.is_available():
        return "cuda"
    raise RuntimeError("QBits: no MI355X/HIP device visible; the gfx950 WOQ path has no CPU fallback")


class PipelineConfig:
    def __init__(self, model_name_or_path="Intel/neural-chat-7b-v3-1", tokenizer_name_or_path=None,
                 hf_endpoint_url=None, hf_access_token=None, device="auto", task="", plugins=None,
                 loading_config=None, optimization_config=None, assistant_model=None, serving_config=None,
                 openai_config=None):
        from ..transformers import (AutoRoundConfig, AwqConfig, GPTQConfig, MixedPrecisionConfig, RtnConfig,
                                    TeqConfig)

        if hf_endpoint_url or openai_config or serving_config:
            raise NotImplementedError("QBits: remote endpoints / OpenAI / vLLM serving are outside the MI355X path")
        self.model_name_or_path = model_name_or_path
        self.tokenizer_name_or_path = tokenizer_name_or_path
        self.hf_access_token = hf_access_token
        self.hf_endpoint_url = None
        self.device = get_device_type() if device == "auto" else device
        self.task = task
        self.plugins = plugins or {}
        self.loading_config = loading_config if loading_config is not None else LoadingModelConfig()
        # same default rule as the reference (:509-510): fp16 on a GPU
        self.optimization_config = optimization_config if optimization_config is not None else \
            MixedPrecisionConfig(dtype="float16")
        allowed = (MixedPrecisionConfig, RtnConfig, AwqConfig, TeqConfig, GPTQConfig, AutoRoundConfig)
        assert type(self.optimization_config) in allowed, \
            "Expect optimization_config be an object of MixedPrecisionConfig, RtnConfig, AwqConfig, TeqConfig, " \
            "GPTQConfig or AutoRoundConfig, got %s." % type(self.optimization_config)
        self.assistant_model = assistant_model
        self.serving_config = None
