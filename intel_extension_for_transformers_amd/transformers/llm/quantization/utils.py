"""Quantisation orchestration for the MI355X backend: checkpoint-format decode, module surgery, model conversion.

Counterpart of intel_extension_for_transformers/transformers/llm/quantization/utils.py:
  unpack_weight                :82-125   INC / optimum / AutoGPTQ on-disk tensors -> int8 [K,N], scales, zeros
  replace_linear/_replace_linear :128-434  swap nn.Linear (and packed checkpoint linears) for QuantizedLinearQBits
  convert_to_quantized_model   :531-702  config -> quantised model
The reference's `_replace_linear` has a cpu branch (:278-295), an xpu branch (:296-354) and raises for anything else
(:355-360); this file is the missing `cuda` (= HIP) branch. The reference delegates the RTN rounding itself to
neural_compressor (external); here RTN runs on the GPU inside qbits.quantize_to_packed_weight
(rule: DESIGN.md "RTN", parity unpinned), and pre-quantised tensors take the exact reference route
unpack_weight -> set_weights_bias -> repack_quantized_weight.
"""
import logging

import torch

from .nn.modules import QuantizedLinearQBits

logger = logging.getLogger(__name__)

_DTYPE_STR = {torch.float32: "fp32", torch.float16: "fp16", torch.bfloat16: "bf16", torch.int8: "int8"}


def convert_dtype_torch2str(dtype):
    """reference utils.py:43-62."""
    if isinstance(dtype, str) or dtype is None:
        return dtype
    if dtype in _DTYPE_STR:
        return _DTYPE_STR[dtype]
    raise AssertionError("Unsupported pytorch dtype {} to str dtype".format(dtype))


def convert_dtype_str2torch(s):
    if isinstance(s, torch.dtype) or s is None:
        return s
    for k, v in _DTYPE_STR.items():
        if v == s:
            return k
    raise AssertionError("Unsupported str dtype {} to torch dtype".format(s))


# ---- a13: on-disk (INC / optimum / AutoGPTQ) tensor format ---------------------------------------------------------
def unpack_weight(qweight, scales, qzeros, q_config):
    """`qweight` int32 [K/n_pack, N] (value j of word i = row n_pack*i + j), `qzeros` int32 [G, N/n_pack] packed along
    N and STORING zp - 1 -> (int8 [K, N] unsigned values, scales, int8 [G, N] zeros un-biased by +1).
    Semantics of reference utils.py:82-125, including the 8-bit conventions (sym: -128 bias :116-119; asym: shift to
    int8 :103-106,121-124). Vectorised with shifts on the tensors' own device."""
    bits = int(q_config.bits)
    sym = bool(q_config.sym)
    n_pack = 32 // bits
    mask = (1 << bits) - 1
    shifts = torch.arange(0, 32, bits, dtype=torch.int32, device=qweight.device)
    wide = torch.int16 if bits == 8 else torch.int8

    zeros = None
    if qzeros is not None:
        z = (qzeros.to(torch.int32).unsqueeze(-1) >> shifts.view(1, 1, n_pack)) & mask  # [G, N/n_pack, n_pack]
        z = z.to(wide)
        if bits == 8:
            z = z.to(torch.int8 if sym else torch.uint8)
        z = z + 1  # the writer stored zp - 1 (utils.py:93-94)
        z = z.reshape(z.shape[0], -1)
        if z.shape != scales.shape:  # N not a multiple of n_pack: drop the padding columns
            z = z[:, :scales.shape[1]]
        if not sym and bits == 8:
            z = (z.to(torch.int32) - 128).to(torch.int8)
        zeros = z.contiguous()

    w = (qweight.to(torch.int32).unsqueeze(1) >> shifts.view(1, n_pack, 1)) & mask  # [K/n_pack, n_pack, N]
    w = w.reshape(-1, qweight.shape[-1]).to(wide)
    if bits == 8:
        if sym:
            w = w - 128
        w = w.to(torch.int8 if sym else torch.uint8)
        if not sym:
            w = (w.to(torch.int32) - 128).to(torch.int8)
    return w.contiguous(), scales.contiguous(), zeros


_AWQ_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)  # nibble i of an AutoAWQ word holds column 8 * word + _AWQ_ORDER[i]


def unpack_awq_gemm(qweight, scales, qzeros):
    """AutoAWQ "GEMM" checkpoint tensors -> (unsigned int8 [K, N], scales, unsigned zeros [G, N]). Packing of that
    writer (4-bit): `qweight` int32 [K, N/8] and `qzeros` int32 [G, N/8], both packed ALONG N, nibble i of word c =
    column 8c + (0,2,4,6,1,3,5,7)[i]; zero points stored as they are (no -1, unlike the GPTQ / INC format that
    `unpack_weight` reads). w = (q - z) * s."""
    shifts = torch.arange(0, 32, 4, dtype=torch.int32, device=qweight.device)
    order = torch.tensor(_AWQ_ORDER, device=qweight.device)

    def along_n(t):
        v = ((t.to(torch.int32).unsqueeze(-1) >> shifts.view(1, 1, 8)) & 15).to(torch.int8)  # [R, N/8, 8]: nibble order
        out = torch.empty_like(v)
        out[:, :, order] = v  # column 8c + order[i] <- nibble i
        return out.reshape(t.shape[0], -1)

    w = along_n(qweight)[:, :scales.shape[1]].contiguous()
    z = along_n(qzeros)[:, :scales.shape[1]].contiguous() if qzeros is not None else None
    return w, scales.contiguous(), z


def pack_awq_gemm(int_weight, zeros):
    """Inverse of unpack_awq_gemm (test / export helper): unsigned [K, N] and [G, N] -> int32 words along N."""
    order = torch.tensor(_AWQ_ORDER, device=int_weight.device)
    shifts = torch.arange(0, 32, 4, dtype=torch.int64, device=int_weight.device)

    def along_n(t):
        r, n = t.shape
        v = t.to(torch.int64).reshape(r, n // 8, 8)[:, :, order]  # nibble i <- column 8c + order[i]
        return _to_i32((v << shifts.view(1, 1, 8)).sum(-1))

    return along_n(int_weight), along_n(zeros)


def pack_weight(int_weight, scales, zeros, bits=4):
    """Inverse of unpack_weight for the save path (what INC's WeightOnlyLinear.pack writes, reference
    modeling_auto.py:128-149): unsigned int8 [K, N] -> qweight int32 [ceil(K/n_pack), N]; unsigned zeros [G, N] ->
    qzeros int32 [G, ceil(N/n_pack)] storing zp - 1; scales -> fp16 ("optimum format")."""
    n_pack = 32 // bits
    mask = (1 << bits) - 1
    k, n = int_weight.shape
    kp = (k + n_pack - 1) // n_pack * n_pack
    w = torch.zeros(kp, n, dtype=torch.int64, device=int_weight.device)
    w[:k] = int_weight.to(torch.int64) & mask
    shifts = torch.arange(0, 32, bits, dtype=torch.int64, device=w.device)
    qweight = (w.view(kp // n_pack, n_pack, n) << shifts.view(1, n_pack, 1)).sum(1)
    qweight = _to_i32(qweight)
    qzeros = None
    if zeros is not None:
        g = zeros.shape[0]
        np_ = (n + n_pack - 1) // n_pack * n_pack
        z = torch.zeros(g, np_, dtype=torch.int64, device=zeros.device)
        z[:, :n] = (zeros.to(torch.int64) - 1) & mask
        qzeros = _to_i32((z.view(g, np_ // n_pack, n_pack) << shifts.view(1, 1, n_pack)).sum(2))
    return qweight, scales.to(torch.float16), qzeros


def _to_i32(t64):
    """low 32 bits of an int64 tensor, reinterpreted as int32."""
    t = t64 & 0xFFFFFFFF
    return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)


# ---- a15: module surgery --------------------------------------------------------------------------------------------
def _is_conv1d(m):
    return type(m).__name__ == "Conv1D" and hasattr(m, "nf")  # HF GPT-2 projection: weight already [K, N] (F10)


def _packed_checkpoint_linear(m):
    """A linear that already carries optimum-format tensors (INC WeightOnlyLinear / AutoGPTQ QuantLinear shells)."""
    return all(hasattr(m, a) for a in ("qweight", "scales")) and not isinstance(m, QuantizedLinearQBits)


def replace_linear(model, modules_to_not_convert=None, current_key_name=None, quantization_config=None, device="cuda",
                   empty_weights=False):
    """reference utils.py:128-162. Swaps every eligible linear; warns if nothing was replaced."""
    if modules_to_not_convert is None:
        modules_to_not_convert = []
    skip = list(getattr(quantization_config, "llm_int8_skip_modules", None) or [])
    modules_to_not_convert = list(dict.fromkeys(list(modules_to_not_convert) + skip))
    model, replaced = _replace_linear(model, modules_to_not_convert, current_key_name, quantization_config,
                                      device=device, empty_weights=empty_weights)
    if not replaced:
        logger.warning("You are loading your model in 8bit or 4bit but no linear modules were found in your model. "
                       "Please double check your model architecture, or submit an issue on github if you think this "
                       "is a bug.")
    return model


def _replace_linear(model, modules_to_not_convert, current_key_name, quantization_config, is_replaced=False,
                    device="cuda", empty_weights=False):
    """Recursive walk (reference utils.py:164-434). Eligible: nn.Linear, HF Conv1D, packed-checkpoint linears; a module
    is skipped when any entry of `modules_to_not_convert` is a substring of its dotted name (:189-201)."""
    cfg = quantization_config
    for name, module in list(model.named_children()):
        key = (current_key_name or []) + [name]
        dotted = ".".join(key)
        eligible = (isinstance(module, torch.nn.Linear) and not isinstance(module, QuantizedLinearQBits)) \
            or _is_conv1d(module) or _packed_checkpoint_linear(module)
        if eligible and not any(s in dotted for s in modules_to_not_convert):
            if _is_conv1d(module):
                in_features, out_features = module.weight.shape[0], module.nf
            elif hasattr(module, "in_features"):
                in_features, out_features = module.in_features, module.out_features
            else:
                in_features, out_features = module.infeatures, module.outfeatures
            has_bias = getattr(module, "bias", None) is not None
            new = QuantizedLinearQBits(in_features, out_features, has_bias, compute_dtype=cfg.compute_dtype,
                                       compress_statistics=False, weight_dtype=cfg.weight_dtype, bits=cfg.bits,
                                       scale_dtype=cfg.scale_dtype, blocksize=cfg.group_size, scheme=cfg.scheme,
                                       device=device, use_optimum_format=_packed_checkpoint_linear(module))
            if not empty_weights:
                bias = module.bias.data if has_bias else None
                if _packed_checkpoint_linear(module):
                    g_idx = getattr(module, "g_idx", None)
                    if getattr(module, "awq_gemm", False):
                        int_w, scales, zeros = unpack_awq_gemm(module.qweight, module.scales,
                                                               getattr(module, "qzeros", None))
                    else:
                        int_w, scales, zeros = unpack_weight(module.qweight, module.scales,
                                                             getattr(module, "qzeros", None), cfg)
                    int_w = int_w[:in_features]
                    new.set_weights_bias(int_w, scales, zeros,
                                         g_idx if g_idx is not None else torch.empty(0, dtype=torch.int32), cfg, bias)
                else:
                    w = module.weight.data
                    w = w.t() if _is_conv1d(module) else w  # -> nn.Linear layout [N, K]
                    new.set_fp_weights_bias(w.to(device), bias)
            new.source_cls = type(module)
            new.requires_grad_(False)
            model._modules[name] = new
            is_replaced = True
        elif len(list(module.children())) > 0:
            _, is_replaced = _replace_linear(module, modules_to_not_convert, key, cfg, is_replaced, device,
                                             empty_weights)
    return model, is_replaced


def convert_to_quantized_model(model, config, device="cuda"):
    """reference utils.py:531-702 for the weight-only configs. RTN: quantise every eligible linear on the device.
    AWQ / TEQ / GPTQ / AutoRound need calibration passes that the reference runs inside neural_compressor (external);
    here those configs are accepted for PRE-QUANTISED checkpoints (optimum-format tensors already on the modules),
    and raise for an fp model instead of silently falling back to RTN."""
    method = getattr(config.quant_method, "value", config.quant_method)
    has_packed = any(_packed_checkpoint_linear(m) for m in model.modules())
    if method != "rtn" and not has_packed:
        raise NotImplementedError(
            "%s calibration is not part of the MI355X hot path (the reference delegates it to neural_compressor); "
            "load a pre-quantised checkpoint or use RtnConfig" % type(config).__name__)
    if str(device) == "cpu":
        raise RuntimeError("QBits: the MI355X backend has no CPU path (device must be 'cuda')")
    if getattr(config, "compute_dtype", None) == "int8":
        # SURVEY §8 a7 / VERDICT r05: say so loudly rather than compute at another precision in silence
        logger.warning("compute_dtype='int8' selects the reference's dynamic u8 activation quantisation cores "
                       "(bestla_weightonly_dispatcher.cpp:131-149); the MI355X path accepts the string but multiplies "
                       "with one fp16-class product per weight (HIGHER precision than the reference's u8 activations): "
                       "outputs follow the fp32 dequantise -> matmul definition, not the reference's int8 numerics.")
    orig_dtype = next((p.dtype for p in model.parameters()), torch.float32)
    model = replace_linear(model, None, None, config, device=device)
    model.to(device)
    model.eval()
    model._woq_orig_dtype = orig_dtype
    return model
