"""MI355X-native (gfx950) int4 weight-only-quantized LLM inference path behind the ITREX API.

Drop-in for ONE hot path of intel/intel-extension-for-transformers: the qbits int4 WOQ linear
(+ RMSNorm / RoPE / SiLU|GeLU between the linears) behind AutoModelForCausalLM + RtnConfig-family
configs. See DESIGN.md for scope and the reference file:line map.
"""
__version__ = "0.1.0"
