"""Weight-only quantization configs with the reference's names, fields, defaults and validators.

Mirrors intel_extension_for_transformers/transformers/utils/config.py: `ITREXQuantizationConfigMixin` (:251),
`RtnConfig` (:794), `GPTQConfig` (:865), `AwqConfig` (:979), `TeqConfig` (:1051), `AutoRoundConfig` (:1118), and
adds the device validator the reference lacks, `post_init_hip()`, next to post_init_cpu / post_init_xpu /
post_init_runtime (:277, :374, :425). `WeightOnlyQuantConfig` (the pre-split name the north-star uses; absent from
the reference snapshot) is an alias of RtnConfig.

The per-algorithm classes are table-driven here: each declares (field, default) pairs; the shared constructor,
dict / JSON round trip and diff-against-defaults live in the mixin.
"""
import copy
import json
import os
from enum import Enum

try:  # HF's mixin gives `isinstance(cfg, QuantizationConfigMixin)` to callers that check for it
    from transformers.utils.quantization_config import QuantizationConfigMixin as _HFBase
except Exception:  # pragma: no cover
    _HFBase = object


class QuantizationMethod(str, Enum):
    GPTQ = "gptq"
    AWQ = "awq"
    RTN = "rtn"
    AUTOROUND = "autoround"
    TEQ = "teq"


_SKIP_DEFAULT = ["lm_head", "transformer.output_layer", "embed_out"]  # config.py:836-837
_TORCH_NAMES = {"torch.float32": "fp32", "torch.float16": "fp16", "torch.bfloat16": "bf16", "torch.int8": "int8"}
_NOT_SERIALIZED = ("tokenizer", "calib_dataloader", "calib_func")


def _dtype_str(v):
    """torch.dtype -> the reference's short string (utils.py convert_dtype_torch2str); strings pass through."""
    return _TORCH_NAMES.get(str(v), v) if v is not None and not isinstance(v, str) else v


class ITREXQuantizationConfigMixin(_HFBase):
    """Shared behaviour of every weight-only config (reference config.py:251-660)."""

    quant_method = None
    _fields = ()  # ((name, default), ...) in the reference's constructor order

    def __init__(self, *args, **kwargs):
        names = [n for n, _ in self._fields]
        if len(args) > len(names):
            raise TypeError("%s takes at most %d positional arguments" % (type(self).__name__, len(names)))
        given = dict(zip(names, args))
        for n, default in self._fields:
            if n in kwargs:
                given[n] = kwargs.pop(n)
            setattr(self, n, copy.deepcopy(given.get(n, default)))
        for n in ("compute_dtype", "scale_dtype", "double_quant_scale_dtype"):
            if hasattr(self, n):
                setattr(self, n, _dtype_str(getattr(self, n)))
        self.quant_method = type(self).quant_method
        self.scheme = "sym" if getattr(self, "sym", True) else "asym"
        self.llm_int8_skip_modules = list(kwargs.get("llm_int8_skip_modules", _SKIP_DEFAULT))
        self.device = kwargs.get("device", "auto")
        self.use_ipex = kwargs.pop("use_ipex", False)
        if not hasattr(self, "use_double_quant"):
            self.use_double_quant = False

    # ---- reference API: update / dict / json -------------------------------------------------------------------
    def update(self, **kwargs):
        """config.py:254-275: set matching attributes, return the rest."""
        unused = {}
        for k, v in kwargs.items():
            if hasattr(self, k):
                setattr(self, k, v)
            else:
                unused[k] = v
        return unused

    def to_dict(self):
        out = {}
        for k, v in self.__dict__.items():
            if k in _NOT_SERIALIZED:
                continue
            out[k] = v.value if isinstance(v, Enum) else copy.deepcopy(v)
        return out

    def to_diff_dict(self):
        """Only what differs from a default-constructed config (config.py:844-862)."""
        base = type(self)().to_dict()
        return {k: v for k, v in self.to_dict().items() if k not in base or base[k] != v}

    def to_json_string(self, use_diff=True):
        return json.dumps(self.to_diff_dict() if use_diff else self.to_dict(), indent=2, sort_keys=True) + "\n"

    def to_json_file(self, json_file_path, use_diff=True):
        with open(json_file_path, "w", encoding="utf-8") as f:
            f.write(self.to_json_string(use_diff=use_diff))

    def __repr__(self):
        return "%s %s" % (type(self).__name__, self.to_json_string(use_diff=False))

    @classmethod
    def from_dict(cls, config_dict, return_unused_kwargs=False, **kwargs):
        d = dict(config_dict)
        d.pop("quant_method", None)
        d.pop("scheme", None)
        names = {n for n, _ in cls._fields} | {"llm_int8_skip_modules", "device", "use_ipex"}
        cfg = cls(**{k: v for k, v in d.items() if k in names})
        for k, v in d.items():
            if k not in names:
                setattr(cfg, k, v)
        unused = cfg.update(**kwargs)
        return (cfg, unused) if return_unused_kwargs else cfg

    @classmethod
    def from_json_file(cls, json_file_path):
        with open(json_file_path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    def save_pretrained(self, save_directory, **kwargs):
        """Writes quantize_config.json next to the weights (config.py:639-641, utility.py:34)."""
        os.makedirs(save_directory, exist_ok=True)
        self.to_json_file(os.path.join(save_directory, "quantize_config.json"), use_diff=False)

    def remove_redundant_parameters(self):
        for k in ("calib_dataloader", "dataset", "calib_func", "calib_iters", "calib_len", "double_quant_scale_dtype",
                  "use_double_quant", "mse_range", "scheme", "tokenizer", "use_ggml", "use_neural_speed", "use_quant",
                  "layer_wise", "blocksize", "nsamples", "max_input_length", "lr", "minmax_lr",
                  "iters", "use_quant_input", "device", "calib_shuffle", "calib_padding", "example_inputs",
                  "excluded_precisions", "op_name_dict", "op_type_dict", "train_dataloader", "train_func",
                  "train_iters", "train_len", "train_padding", "train_shuffle", "train_batch_size"):
            self.__dict__.pop(k, None)
        # the reference drops `static_groups` too (config.py:576); a TRUE value is kept here: together with desc_act
        # it decides whether a saved checkpoint carries an activation order, and a reload must read it the same way
        if not self.__dict__.get("static_groups", False):
            self.__dict__.pop("static_groups", None)

    # ---- validators --------------------------------------------------------------------------------------------
    def _common_checks(self):
        if not isinstance(self.use_double_quant, bool):
            raise ValueError("use_double_quant must be a boolean")
        if not isinstance(self.group_size, int):
            raise ValueError("group_size must be a int")
        if not isinstance(self.scheme, str):
            raise ValueError("scheme must be a string")

    def post_init_cpu(self):
        """config.py:277-372."""
        if self.compute_dtype is None:
            self.compute_dtype = "fp32"
        elif self.compute_dtype not in ("fp32", "bf16", "int8"):
            raise ValueError("compute_dtype must be 'fp32', 'bf16', 'int8'.")
        if self.bits is None:
            self.bits = 4
        elif self.bits not in (4, 8):
            raise ValueError("Only support quantization to [4, 8] bits but found %s" % self.bits)
        self.weight_dtype = {"int4": "int4_clip", "fp4": "fp4_e2m1"}.get(self.weight_dtype, self.weight_dtype)
        if self.bits == 4 and self.weight_dtype not in ("int4_clip", "nf4", "fp4_e2m1"):
            self.weight_dtype = "int4_clip"
        if self.bits == 8 and self.weight_dtype not in ("int8", "fp8_e5m2", "fp8_e4m3"):
            self.weight_dtype = "int8"
        if self.scale_dtype is None:
            self.scale_dtype = "fp32"
        elif self.scale_dtype not in ("fp32", "fp8_e8m0", "bf16"):
            raise ValueError("scale_dtype must be a string in 'fp32', 'fp8_e8m0', 'bf16' "
                             "and fp8_e8m0 only used for weight_dtype 'fp8_e5m2', 'fp8_e4m3'")
        self._common_checks()
        floaty = self.weight_dtype.startswith("fp") or self.weight_dtype.startswith("nf")
        if self.scheme == "asym" and ((self.compute_dtype == "int8" and self.weight_dtype == "int8") or floaty
                                      or self.scale_dtype != "fp32"):
            raise ValueError("WeightOnlyQuantization doesn't support asym with compute_dtype int8 or weight_dtype "
                             "float or scale_dtype non-fp32 now, please use sym scheme")
        self.use_neural_speed = False

    def post_init_hip(self):
        """MI355X validator (the device branch the reference lacks, utils.py:355-360 raises for anything but
        cpu/xpu). int4 (bits = 4) or int8 (bits = 8) weights — the reference's `bits in {4, 8}` (config.py:277-372);
        fp32 | bf16 | fp16 scales; any group size that is -1 or a multiple of 32; sym or asym with any scale type
        (the HIP kernels apply zero points exactly)."""
        if self.compute_dtype is None:
            self.compute_dtype = "fp32"
        elif self.compute_dtype not in ("fp32", "bf16", "fp16", "int8"):
            raise ValueError("compute_dtype must be 'fp32', 'bf16', 'fp16' or 'int8'.")
        if self.bits is None:
            self.bits = 4
        elif self.bits not in (4, 8):
            raise ValueError("Only support quantization to [4, 8] bits but found %s" % self.bits)
        fp8 = self.weight_dtype in ("fp8", "fp8_e4m3", "fp8_e5m2")
        if fp8:
            # 8-bit float weights (reference config.py:298-299,313: "fp8" -> "fp8_e4m3", bits = 8): symmetric only,
            # fp32 or power-of-two (fp8_e8m0) scales; generic kernel, off the fused engine
            self.bits = 8
            if self.weight_dtype == "fp8":
                self.weight_dtype = "fp8_e4m3"
            if not self.sym:
                raise ValueError("asym quantization is not supported with %s weights" % self.weight_dtype)
            if self.scale_dtype == "fp8":
                self.scale_dtype = "fp8_e8m0"
            if self.scale_dtype not in (None, "fp32", "fp8_e8m0"):
                raise ValueError("scale_dtype must be 'fp32' or 'fp8_e8m0' with fp8 weights")
        elif self.bits == 8:
            if self.weight_dtype in (None, "int8"):
                self.weight_dtype = "int8"
            else:
                raise ValueError("weight_dtype must be 'int8', 'fp8_e4m3' or 'fp8_e5m2' for bits=8 on the MI355X path, "
                                 "got %s" % self.weight_dtype)
        elif self.weight_dtype in (None, "int4", "int4_fullrange"):
            self.weight_dtype = "int4_clip"
        elif self.weight_dtype in ("nf4", "fp4", "fp4_e2m1", "fp4_e2m1_bnb"):
            # 4-bit table types: symmetric only, like the reference (config.py:360-370: asym is illegal with float
            # weights); they run on the generic kernel (functional, untuned) and stay off the fused engine
            if self.weight_dtype == "fp4":
                self.weight_dtype = "fp4_e2m1"
            if not self.sym:
                raise ValueError("asym quantization is not supported with %s weights" % self.weight_dtype)
        elif self.weight_dtype != "int4_clip":
            raise ValueError("weight_dtype must be 'int4' / 'int4_clip' / 'nf4' / 'fp4_e2m1' / 'fp4_e2m1_bnb' (bits=4) "
                             "or 'int8' (bits=8) on the MI355X path, got %s" % self.weight_dtype)
        if self.scale_dtype is None:
            self.scale_dtype = "fp32"
        elif self.scale_dtype not in ("fp32", "bf16", "fp16") and not (fp8 and self.scale_dtype == "fp8_e8m0"):
            raise ValueError("scale_dtype must be a string in 'fp32', 'bf16', 'fp16' ('fp8_e8m0' with fp8 weights)")
        self._common_checks()
        if self.group_size != -1 and (self.group_size <= 0 or self.group_size % 32 != 0):
            raise ValueError("group_size must be -1 or a positive multiple of 32")
        self.use_neural_speed = False

    def post_init_xpu(self):
        """config.py:374-423 (kept for API parity; the Intel-GPU backend itself is not part of this build)."""
        if self.compute_dtype is None:
            self.compute_dtype = "fp16"
        elif self.compute_dtype != "fp16":
            raise ValueError("compute_dtype must be 'fp16'.")
        if self.bits is None:
            self.bits = 4
        elif self.bits != 4:
            raise ValueError("Only support quantization to [4] bits but found %s" % self.bits)
        if self.weight_dtype in (None, "int4"):
            self.weight_dtype = "int4_fullrange"
        elif self.weight_dtype != "int4_fullrange":
            raise ValueError("weight_dtype must be a string in 'int4_fullrange', but get %s." % self.weight_dtype)
        if self.scale_dtype is None:
            self.scale_dtype = "fp16"
        elif self.scale_dtype != "fp16":
            raise ValueError("scale_dtype must be a string in 'fp16'")
        self._common_checks()
        if self.scheme != "sym":
            raise ValueError("scheme: %s is not support, only support 'sym' now!" % self.scheme)
        self.use_neural_speed = False

    def post_init_runtime(self):
        """config.py:425-538: the Neural-Speed runtime's validator with its documented fall-backs."""
        if self.compute_dtype is None:
            self.compute_dtype = "fp32"
        elif self.compute_dtype not in ("fp32", "fp16", "bf16", "int8"):
            raise ValueError("compute_dtype must be in ['fp32', 'fp16', 'bf16', 'int8'].")
        if self.bits is None:
            self.bits = 4
        elif self.bits not in (4, 8):
            raise ValueError("Only support quantization to [4, 8] bits but found %s" % self.bits)
        alias = {None: "int4", "int4_clip": "int4", "int4_fullrange": "int4", "fp8": "fp8_e4m3", "fp4": "fp4_e2m1"}
        if self.weight_dtype in alias:
            self.weight_dtype = alias[self.weight_dtype]
        elif self.weight_dtype not in ("int4", "int8", "fp8_e5m2", "fp8_e4m3", "fp4_e2m1", "nf4"):
            raise ValueError("unsupported weight_dtype %s" % self.weight_dtype)
        if self.bits == 4 and self.weight_dtype not in ("int4", "nf4", "fp4_e2m1"):
            self.weight_dtype = "int4"
        if self.bits == 8 and self.weight_dtype not in ("int8", "fp8_e5m2", "fp8_e4m3"):
            self.weight_dtype = "int8"
        if self.scale_dtype is None:
            self.scale_dtype = "fp32"
        elif self.scale_dtype not in ("fp32", "bf16", "fp8"):
            raise ValueError("scale_dtype must be in ['fp32', 'bf16', 'fp8'].")
        if self.group_size not in (-1, 32, 128):
            raise ValueError("group_size must be an integer in [-1, 32, 128].")
        if self.weight_dtype[:3] in ("fp8", "fp4", "nf4"):
            if self.compute_dtype == "int8":
                self.compute_dtype = "fp32"
            if self.scheme == "asym":
                self.scheme = "sym"
            if self.scale_dtype == "fp8" and self.weight_dtype[:3] != "fp8":
                self.scale_dtype = "fp32"
            if self.weight_dtype[:3] == "fp8" and self.scale_dtype not in ("fp8", "fp32"):
                self.scale_dtype = "fp8"
        self.use_neural_speed = True


_TAIL = (("use_ggml", False), ("use_quant", True), ("use_neural_speed", False))


class MixedPrecisionConfig:
    """Reference transformers/utils/config.py:59-66: no weight quantisation, just the model dtype. NeuralChat's
    default `optimization_config` (neural_chat/config.py:509-510)."""

    def __init__(self, dtype="bfloat16"):
        self.dtype = dtype


class RtnConfig(ITREXQuantizationConfigMixin):
    """Round-to-nearest (config.py:794-842). Defaults: bits 4, group_size 32, sym."""

    quant_method = QuantizationMethod.RTN
    _fields = (("bits", 4), ("group_size", 32), ("group_dim", 1), ("compute_dtype", None), ("weight_dtype", None),
               ("scale_dtype", None), ("use_full_range", False), ("mse_range", False), ("use_double_quant", False),
               ("double_quant_dtype", "int"), ("double_quant_bits", 8), ("double_quant_use_sym", False),
               ("double_quant_group_size", 256), ("sym", True), ("layer_wise", False)) + _TAIL


class GPTQConfig(ITREXQuantizationConfigMixin):
    """config.py:865-977."""

    quant_method = QuantizationMethod.GPTQ
    _fields = (("bits", 4), ("tokenizer", None), ("dataset", "NeelNanda/pile-10k"), ("batch_size", 8),
               ("group_size", 32), ("compute_dtype", None), ("weight_dtype", None), ("scale_dtype", None),
               ("use_double_quant", False), ("double_quant_scale_dtype", None), ("sym", True), ("blocksize", 128),
               ("damp_percent", 0.1), ("desc_act", False), ("n_samples", 128), ("seq_len", 2048),
               ("static_groups", False), ("use_mse_search", False), ("true_sequential", False),
               ("layer_wise", False)) + _TAIL


class AwqConfig(ITREXQuantizationConfigMixin):
    """config.py:979-1049. `zero_point=True` means asym."""

    quant_method = QuantizationMethod.AWQ
    _fields = (("bits", 8), ("tokenizer", None), ("dataset", "NeelNanda/pile-10k"), ("group_size", 32),
               ("compute_dtype", None), ("weight_dtype", None), ("scale_dtype", None), ("layer_wise", False),
               ("n_samples", 128), ("seq_len", 2048), ("auto_scale", True), ("auto_clip", True),
               ("use_double_quant", False), ("double_quant_scale_dtype", None), ("zero_point", True)) + _TAIL

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.sym = not self.zero_point
        self.scheme = "sym" if self.sym else "asym"


class TeqConfig(ITREXQuantizationConfigMixin):
    """config.py:1051-1116."""

    quant_method = QuantizationMethod.TEQ
    _fields = (("bits", 8), ("tokenizer", None), ("dataset", "NeelNanda/pile-10k"), ("group_size", 32),
               ("compute_dtype", None), ("weight_dtype", None), ("scale_dtype", None), ("layer_wise", False),
               ("absorb_to_layer", {}), ("n_samples", 128), ("seq_len", 2048), ("use_double_quant", False),
               ("double_quant_scale_dtype", None), ("sym", True)) + _TAIL


class AutoRoundConfig(ITREXQuantizationConfigMixin):
    """config.py:1118-1200."""

    quant_method = QuantizationMethod.AUTOROUND
    _fields = (("bits", 4), ("tokenizer", None), ("dataset", "NeelNanda/pile-10k"), ("group_size", 128),
               ("compute_dtype", None), ("weight_dtype", None), ("scale_dtype", None), ("use_double_quant", False),
               ("double_quant_scale_dtype", None), ("sym", False), ("lr", None), ("minmax_lr", None),
               ("disable_quanted_input", True), ("n_samples", 128), ("seq_len", 2048), ("iters", 200),
               ("quant_lm_head", False), ("layer_wise", False)) + _TAIL


# the pre-split name used by BASELINE.json's north star (not present in the reference snapshot)
WeightOnlyQuantConfig = RtnConfig
