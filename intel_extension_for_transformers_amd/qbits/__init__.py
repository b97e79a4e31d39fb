"""`qbits` operator module for MI355X — same function names, argument order and meaning as the
reference's pybind module `qbits_py` (intel_extension_for_transformers/qbits/qbits.cpp:192-206,
re-exported by qbits/__init__.py:19-21), implemented as thin ctypes calls into libwoq_hip.so.

All tensors are torch CUDA(=HIP) tensors; launches go on torch.cuda.current_stream(); nothing here
allocates on the hot call (`woq_linear` writes the caller's pre-allocated `output` in place exactly like
qbits.cpp:113-140). Errors raise RuntimeError with the library's "QBits: ..." text.
"""
import ctypes

import torch

from .. import _lib as L

__all__ = [
    "woq_linear", "repack_quantized_weight", "quantize_to_packed_weight", "dequantize_packed_weight",
    "acquire_packed_weight_info", "get_packed_weight_size", "set_woq_workspace", "set_qbits_threads",
    "check_isa_supported", "check_torch_compatibility", "rmsnorm", "rope", "silu_mul", "gelu",
]


import os as _os

_VERBOSE = _os.environ.get("QBITS_VERBOSE") is not None  # dispatcher_utils.hpp:51-58


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _blocksize_ok(k, blocksize):
    return blocksize == -1 or blocksize >= k or blocksize % 32 == 0


def _wtype(weight_type):
    if weight_type not in L.WEIGHT_TYPES:
        # reference text: bestla_packq_impl.cpp "unsupported bestla packq config"
        raise RuntimeError("Qbits: unsupported bestla packq config, weight_type: %s (MI355X path: int4_clip, int3_clip, int2_clip, int8, nf4, fp4_e2m1, fp4_e2m1_bnb, fp8_e4m3, fp8_e5m2)"
                           % weight_type)
    return L.WEIGHT_TYPES[weight_type]


def _stype(scale_type):
    if scale_type not in L.SCALE_TYPES:
        raise RuntimeError("QBits: unsupported scale_type %s (fp32 | bf16 | fp16; fp8_e8m0 with fp8 weights)" % scale_type)
    return L.SCALE_TYPES[scale_type]


def _ctype(compute_type):
    if compute_type not in L.COMPUTE_TYPES:
        raise RuntimeError("Qbits: unsupported bestla_config, compute_type: %s" % compute_type)
    return L.COMPUTE_TYPES[compute_type]


def header_of(packed):
    """Cached host copy of the blob header (the reference re-parses it on every call,
    bestla_weightonly_dispatcher.cpp:335; here it is read once per tensor storage)."""
    key = (packed.data_ptr(), packed.numel())
    cached = getattr(packed, "_woq_hdr", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    if not packed.is_cuda:
        raise RuntimeError("QBits: packed weight must live on the HIP device")
    hdr = L.BlobHeader()
    L.check(L.lib().woq_read_header(_ptr(packed), ctypes.byref(hdr), L.stream_ptr()))
    try:
        packed._woq_hdr = (key, hdr)
    except Exception:  # pragma: no cover
        pass
    return hdr


def get_packed_weight_size(k, n, weight_type, scale_type, compute_type, asym, blocksize, act_shuf):
    """qbits.cpp:79-88."""
    wt = _wtype(weight_type)
    if asym and wt in (L.W_NF4, L.W_FP4_E2M1, L.W_FP4_E2M1_BNB, L.W_FP8_E4M3, L.W_FP8_E5M2):
        raise RuntimeError("QBits: float weight types (nf4 / fp4 / fp8) are symmetric: asym is not supported")
    if _stype(scale_type) == L.S_FP8_E8M0 and wt not in (L.W_FP8_E4M3, L.W_FP8_E5M2):
        # the reference's validity matrix: qbits_ut/test_weightonly.py:24-25
        raise RuntimeError("QBits: fp8_e8m0 scales are only used with fp8_e4m3 / fp8_e5m2 weights")
    size = L.lib().woq_packed_weight_size(k, n, blocksize, _wtype(weight_type), _stype(scale_type), int(asym),
                                          int(act_shuf))
    if size == 0:
        raise RuntimeError("QBits: unsupported blocksize %d (must be -1 or a multiple of 32)" % blocksize)
    return size


def repack_quantized_weight(qweight, scale, zp, g_idx, weight_type, scale_type, compute_type, asym, blocksize):
    """qbits.cpp:61-77: int8 [K,N] (signed int4 values, or full int8 for weight_type "int8") + fp32 scales [G,N] +
    int8 zp [G,N] + int32 g_idx -> blob.
    Empty `zp` / `g_idx` tensors mean "absent", as in the reference (qbits.cpp:67, modules.py:233-237)."""
    L.require_gpu()
    dev = qweight.device if qweight.is_cuda else torch.device("cuda", torch.cuda.current_device())
    qweight = qweight.to(dev, torch.int8).contiguous()
    scale = scale.to(dev, torch.float32).contiguous()
    k, n = qweight.shape
    has_zp = bool(asym) and zp is not None and zp.numel() != 0
    has_idx = g_idx is not None and g_idx.numel() != 0
    zp_d = zp.to(dev, torch.int8).contiguous() if has_zp else None
    idx_d = g_idx.to(dev, torch.int32).contiguous() if has_idx else None
    size = get_packed_weight_size(k, n, weight_type, scale_type, compute_type, has_zp, blocksize, has_idx)
    out = torch.empty(size, dtype=torch.int8, device=dev)
    L.check(L.lib().woq_repack_quantized_weight(_ptr(qweight), _ptr(scale), _ptr(zp_d), _ptr(idx_d), k, n, blocksize,
                                                _wtype(weight_type), _stype(scale_type), _ctype(compute_type),
                                                _ptr(out), size, L.stream_ptr()))
    header_of(out)
    return out


def quantize_to_packed_weight(fp32_weight, transpose, blocksize, compute_type, weight_type, scale_type, asym):
    """qbits.cpp:90-100: RTN of an fp32 weight ([N,K] when transpose) into a blob."""
    L.require_gpu()
    dev = fp32_weight.device if fp32_weight.is_cuda else torch.device("cuda", torch.cuda.current_device())
    w = fp32_weight.to(dev, torch.float32).contiguous()
    k, n = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    size = get_packed_weight_size(k, n, weight_type, scale_type, compute_type, asym, blocksize, False)
    out = torch.empty(size, dtype=torch.int8, device=dev)
    L.check(L.lib().woq_quantize_to_packed_weight(_ptr(w), int(transpose), k, n, blocksize, _wtype(weight_type),
                                                  _stype(scale_type), _ctype(compute_type), int(asym), _ptr(out),
                                                  size, L.stream_ptr()))
    header_of(out)
    return out


def dequantize_packed_weight(compressed_weight, dequantize_weight, transpose, compute_type, weight_type, scale_type):
    """qbits.cpp:102-111: in place into the caller-allocated fp32 tensor."""
    hdr = header_of(compressed_weight)
    want = (hdr.N, hdr.K) if transpose else (hdr.K, hdr.N)
    if (tuple(dequantize_weight.shape) != want or dequantize_weight.dtype != torch.float32
            or not dequantize_weight.is_contiguous()):
        raise RuntimeError("QBits: dequantize output must be contiguous fp32 of shape %s" % (want,))
    L.check(L.lib().woq_dequantize_packed_weight(_ptr(compressed_weight), ctypes.byref(hdr), _ptr(dequantize_weight),
                                                 int(transpose), L.stream_ptr()))


def woq_linear(activation, weight, bias, output, compute_type, weight_type, scale_type, asym):
    """qbits.cpp:113-140: output[M,N] = activation[M,K] @ W_deq (+ bias), written in place.
    `bias` empty tensor = none; a non-fp32 bias is converted like qbits.cpp:119-123."""
    hdr = header_of(weight)
    if activation.dim() != 2 or output.dim() != 2:
        raise RuntimeError("QBits: woq_linear expects 2-D activation and output")
    if activation.stride(1) != 1 or activation.stride(0) < activation.shape[1]:  # row-strided views go down as is
        activation = activation.contiguous()
    m, k = activation.shape
    if k != hdr.K or output.shape[0] != m or output.shape[1] != hdr.N:
        raise RuntimeError("QBits: woq_linear shape mismatch: act %s, weight K=%d N=%d, out %s"
                           % (tuple(activation.shape), hdr.K, hdr.N, tuple(output.shape)))
    b = None
    if bias is not None and bias.numel() != 0:
        b = bias if bias.dtype == torch.float32 else bias.float()
        b = b.to(activation.device).contiguous()
    verbose = _VERBOSE  # QBITS_VERBOSE, read once at import like the reference's static env_initer
    if verbose:
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
    L.check(L.lib().woq_linear(_ptr(activation), L.torch_dtype_code(activation.dtype), activation.stride(0),
                               _ptr(weight), ctypes.byref(hdr), _ptr(b), _ptr(output),
                               L.torch_dtype_code(output.dtype), output.stride(0), m, L.stream_ptr()))
    if verbose:  # the reference's per-call line (bestla_weightonly_dispatcher.cpp:180-188); the time is the DEVICE time
        t1.record()  # of this call between two events on its stream (the call itself is asynchronous)
        t1.synchronize()
        print("QBits linear verbose\nm:%d n:%d k:%d weight_type:%s compute_type:%s blocksize:%d src_type:%s dst_type:%s "
              "execute time:%.4fms" % (m, hdr.N, hdr.K, weight_type, compute_type, hdr.group,
                                       str(activation.dtype).replace("torch.", ""),
                                       str(output.dtype).replace("torch.", ""), t0.elapsed_time(t1)), flush=True)


def _ascii(s):
    return torch.tensor([ord(c) for c in s], dtype=torch.int32)


def acquire_packed_weight_info(packw, acquire_type):
    """qbits.cpp:165-167 -> bestla_packq_impl.cpp:152-204. Scalars come back as int64[1], strings as int32
    ASCII arrays, tensors as dense copies — the reference's conventions (:176-198)."""
    hdr = header_of(packw)
    t = int(acquire_type)
    one = lambda v: torch.tensor([int(v)], dtype=torch.int64)  # noqa: E731
    if t == 0:
        return one(hdr.total_bytes)
    if t == 1:
        return one(hdr.group)
    if t == 2:
        return one(hdr.K)
    if t == 3:
        return one(hdr.N)
    if t == 4:
        return one(1 if hdr.off_shuffle else 0)
    if t == 11:
        return one(1 if hdr.off_zp else 0)
    if t == 6:
        # int3_clip / int2_clip blobs are int4 storage with a tag (include/woq_blob.h narrow_bits)
        return _ascii({3: "int3_clip", 2: "int2_clip"}.get(hdr.narrow_bits) or L.WEIGHT_NAMES[hdr.weight_type])
    if t == 7:
        return _ascii(L.COMPUTE_NAMES[hdr.compute_type])
    if t == 8:
        return _ascii("fp8_e8m0" if hdr.flags & L.FLAG_SCALE_E8M0 else L.SCALE_NAMES[hdr.scale_type])
    if t == 5:
        if not hdr.off_shuffle:
            raise RuntimeError("QBits: not pack g_idx tensor.")
        out = torch.empty(hdr.K, dtype=torch.int32, device=packw.device)
    elif t == 9:
        out = torch.empty(hdr.n_groups, hdr.N, dtype=torch.float32, device=packw.device)
    elif t == 10:
        if not hdr.off_zp:
            raise RuntimeError("QBits: not pack zero-point tensor.")
        out = torch.empty(hdr.n_groups, hdr.N, dtype=torch.int8, device=packw.device)
    else:
        raise RuntimeError("QBits: unsupported acquire_type")
    L.check(L.lib().woq_blob_extract(_ptr(packw), ctypes.byref(hdr), t, _ptr(out), L.stream_ptr()))
    return out


_workspace = None  # the tensor whose memory the library points into (it must outlive every call: kept here)


def set_woq_workspace(workspace):
    """qbits.cpp:142-144 -> bestla_weightonly_dispatcher.cpp:394-397: a caller-owned tensor whose memory the library
    uses as scratch from now on (a raw pointer; this module keeps the tensor alive). With it, the calls that need
    scratch — int8 weights, 16-bit activations at M <= 8, the M > 8 GEMM's packed activations — allocate nothing and
    can be captured into a graph; without it (or when it is too small for a call) they use stream-ordered allocation.
    `None` or an empty tensor removes it. Like the reference's, one process-wide workspace: not for concurrent use
    from several threads."""
    global _workspace
    if workspace is None or workspace.numel() == 0:
        L.check(L.lib().woq_set_workspace(None, 0))
        _workspace = None
        return None
    if not workspace.is_cuda or not workspace.is_contiguous():
        raise RuntimeError("QBits: the workspace must be a contiguous tensor on the HIP device")
    L.check(L.lib().woq_set_workspace(_ptr(workspace), workspace.numel() * workspace.element_size()))
    _workspace = workspace
    return None


def set_qbits_threads(thread_num):
    """qbits.cpp:146. CPU thread-pool size has no meaning on the GPU path."""
    return None


def check_isa_supported(isa):
    """qbits.cpp:173-180 answered for this device: the x86 ISA names are all False; 'GFX950'/'MFMA' True on MI355X."""
    if isa in ("GFX950", "MFMA", "CDNA4"):
        return torch.cuda.is_available() and L.lib().woq_device_count() > 0
    return False


def check_torch_compatibility(version):
    """qbits.cpp:182-190. The C ABI carries no torch types, so any torch that can hand out device pointers works."""
    return True


# ---- ops between the linears (no reference boundary exists for these: they replace HF module forwards) ----
def rmsnorm(x, weight, eps, out=None):
    x = x.contiguous()
    d = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().woq_rmsnorm(_ptr(x), L.torch_dtype_code(x.dtype), _ptr(weight.float().contiguous()), float(eps),
                                x.numel() // d, d, _ptr(out), L.torch_dtype_code(out.dtype), L.stream_ptr()))
    return out


def rope(x, pos, cos, sin):
    """In place on x [tokens, heads, D]; cos/sin fp32 [max_pos, D/2]; pos int32 [tokens] (device)."""
    if not x.is_contiguous():
        raise RuntimeError("QBits: rope needs a contiguous [tokens, heads, D] tensor")
    t, h, d = x.shape
    L.check(L.lib().woq_rope(_ptr(x), L.torch_dtype_code(x.dtype), _ptr(pos), _ptr(cos), _ptr(sin), t, h, d,
                             L.stream_ptr()))
    return x


def silu_mul(gate, up, out=None):
    gate, up = gate.contiguous(), up.contiguous()
    out = torch.empty_like(gate) if out is None else out
    L.check(L.lib().woq_silu_mul(_ptr(gate), _ptr(up), L.torch_dtype_code(gate.dtype), gate.numel(), _ptr(out),
                                 L.stream_ptr()))
    return out


def gelu(x, approximate="tanh", out=None):
    x = x.contiguous()
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().woq_gelu(_ptr(x), L.torch_dtype_code(x.dtype), x.numel(), 1 if approximate == "tanh" else 0,
                             _ptr(out), L.stream_ptr()))
    return out
