"""`intel_extension_for_transformers.transformers` surface for the int4 weight-only path, MI355X backend
(reference exports: transformers/__init__.py:28-42)."""
from .utils.config import (AutoRoundConfig, AwqConfig, GPTQConfig, MixedPrecisionConfig, RtnConfig,  # noqa: F401
                           TeqConfig, WeightOnlyQuantConfig)


def __getattr__(name):  # the model classes import torch + HF transformers: load them on first use
    if name in ("AutoModelForCausalLM", "AutoModel", "AutoModelForSeq2SeqLM"):
        from .modeling import modeling_auto

        return getattr(modeling_auto, name)
    raise AttributeError(name)


__all__ = ["AutoModelForCausalLM", "AutoModel", "AutoModelForSeq2SeqLM", "RtnConfig", "AwqConfig", "TeqConfig",
           "GPTQConfig", "AutoRoundConfig", "WeightOnlyQuantConfig", "MixedPrecisionConfig"]
