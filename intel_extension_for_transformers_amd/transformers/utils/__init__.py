from .config import (AwqConfig, AutoRoundConfig, GPTQConfig, ITREXQuantizationConfigMixin, QuantizationMethod,
                     RtnConfig, TeqConfig, WeightOnlyQuantConfig)

__all__ = ["RtnConfig", "AwqConfig", "TeqConfig", "GPTQConfig", "AutoRoundConfig", "WeightOnlyQuantConfig",
           "ITREXQuantizationConfigMixin", "QuantizationMethod"]
