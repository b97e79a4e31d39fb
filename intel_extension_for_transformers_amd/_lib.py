"""ctypes binding of libwoq_hip.so (C ABI declared in include/woq_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError with
the library's "QBits: ..." message is raised — the same exception type the reference's TORCH_CHECKs
surface as (qbits/dispatcher/src/bestla_weightonly_dispatcher.cpp:289,368).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WOQ_HIP_LIB") or os.path.join(_HERE, "libwoq_hip.so")  # env: A/B builds in development

F32, BF16, F16, FP8_E4M3 = 0, 1, 2, 3
W_INT4_CLIP, W_INT8, W_NF4, W_FP4_E2M1, W_FP4_E2M1_BNB, W_INT3_CLIP, W_INT2_CLIP = 0, 1, 2, 3, 4, 5, 6
W_FP8_E4M3, W_FP8_E5M2 = 7, 8
S_FP8_E8M0 = 4       # pack-time scale type for fp8 weights: stored as bf16, header flag FLAG_SCALE_E8M0
FLAG_SCALE_E8M0 = 4
C_FP32, C_BF16, C_INT8, C_FP16 = 0, 1, 2, 3
HEADER_BYTES = 256

WEIGHT_TYPES = {"int4_clip": W_INT4_CLIP, "int4": W_INT4_CLIP, "int8": W_INT8, "nf4": W_NF4, "fp4_e2m1": W_FP4_E2M1,
                "fp4": W_FP4_E2M1, "fp4_e2m1_bnb": W_FP4_E2M1_BNB, "int3_clip": W_INT3_CLIP, "int2_clip": W_INT2_CLIP,
                "fp8_e4m3": W_FP8_E4M3, "fp8": W_FP8_E4M3, "fp8_e5m2": W_FP8_E5M2}
SCALE_TYPES = {"fp32": F32, "bf16": BF16, "fp16": F16, "fp8_e8m0": S_FP8_E8M0}
COMPUTE_TYPES = {"fp32": C_FP32, "bf16": C_BF16, "int8": C_INT8, "fp16": C_FP16}
SCALE_NAMES = {F32: "fp32", BF16: "bf16", F16: "fp16"}
COMPUTE_NAMES = {v: k for k, v in COMPUTE_TYPES.items()}
WEIGHT_NAMES = {W_INT4_CLIP: "int4_clip", W_INT8: "int8", W_NF4: "nf4", W_FP4_E2M1: "fp4_e2m1",
                W_FP4_E2M1_BNB: "fp4_e2m1_bnb", W_FP8_E4M3: "fp8_e4m3", W_FP8_E5M2: "fp8_e5m2"}


class BlobHeader(ctypes.Structure):
    """struct woq_blob_header (include/woq_blob.h)."""

    _fields_ = [
        ("magic", ctypes.c_uint32), ("version", ctypes.c_uint32), ("total_bytes", ctypes.c_uint64),
        ("K", ctypes.c_int32), ("N", ctypes.c_int32), ("group", ctypes.c_int32), ("Kpad", ctypes.c_int32),
        ("Npad", ctypes.c_int32), ("n_groups", ctypes.c_int32),
        ("weight_type", ctypes.c_uint32), ("scale_type", ctypes.c_uint32), ("compute_type", ctypes.c_uint32),
        ("flags", ctypes.c_uint32), ("scale_mode", ctypes.c_uint32), ("narrow_bits", ctypes.c_uint32),
        ("off_q", ctypes.c_uint64), ("off_scale", ctypes.c_uint64), ("off_zp", ctypes.c_uint64),
        ("off_shuffle", ctypes.c_uint64), ("pad", ctypes.c_uint8 * (HEADER_BYTES - 96)),
    ]


assert ctypes.sizeof(BlobHeader) == HEADER_BYTES


class EngineConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("hidden", "inter", "heads", "kv_heads", "head_dim", "layers", "vocab",
                                              "max_ctx")] + [("rms_eps", ctypes.c_float), ("rope_theta", ctypes.c_float),
                                                             ("tp_rank", ctypes.c_int32), ("tp_size", ctypes.c_int32),
                                                             ("kv_dtype", ctypes.c_int32),
                                                             ("reserved", ctypes.c_int32 * 3)]


class LayerWeights(ctypes.Structure):
    _fields_ = [("qkv_blob", ctypes.c_void_p), ("o_blob", ctypes.c_void_p), ("gate_up_blob", ctypes.c_void_p),
                ("down_blob", ctypes.c_void_p), ("ln1", ctypes.c_void_p), ("ln2", ctypes.c_void_p),
                ("qkv_hdr", BlobHeader), ("o_hdr", BlobHeader), ("gate_up_hdr", BlobHeader), ("down_hdr", BlobHeader)]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

# every symbol include/woq_hip.h declares (tests check the .so exports all of them): ABI version 4, the frozen boundary
EXPORTS = [
    "woq_last_error", "woq_abi_version", "woq_device_count", "woq_packed_weight_size",
    "woq_repack_quantized_weight", "woq_quantize_to_packed_weight", "woq_dequantize_packed_weight",
    "woq_read_header", "woq_blob_extract", "woq_linear", "woq_rmsnorm", "woq_rope", "woq_silu_mul", "woq_gelu",
    "woq_engine_create", "woq_engine_destroy", "woq_engine_set_layer", "woq_engine_set_head",
    "woq_engine_bind_io", "woq_engine_token_ptr", "woq_engine_pos_ptr", "woq_engine_logits_ptr", "woq_engine_hidden_ptr",
    "woq_engine_step", "woq_engine_capture", "woq_engine_replay", "woq_engine_steps", "woq_engine_set_allreduce",
    "woq_engine_phase", "woq_engine_prefill", "woq_engine_prefill_logits_ptr", "woq_engine_kv_cache_ptr",
    "woq_engine_set_attn_splits", "woq_engine_attn_splits", "woq_engine_set_attn_grouped", "woq_engine_attn_grouped",
    "woq_engine_set_tp_options", "woq_engine_set_fuse_attn", "woq_engine_fuse_attn", "woq_engine_status",
    "woq_engine_clear_status", "woq_comm_create", "woq_comm_handle", "woq_comm_connect", "woq_comm_allreduce_f32",
    "woq_comm_status", "woq_comm_set_timeout_ms", "woq_comm_destroy", "woq_engine_set_comm", "woq_set_workspace",
    "woq_engine_uses_xq", "woq_engine_token_log_ptr", "woq_table_digit_planes",
]
# include/woq_hip_experimental.h: measurement hooks and lab switches, outside WOQ_ABI_VERSION
EXPERIMENTAL_EXPORTS = [
    "woq_engine_set_attn_chunk", "woq_engine_attn_chunk", "woq_engine_time_gemv", "woq_engine_time_gemv_mask",
    "woq_engine_time_twin", "woq_engine_set_time_eager", "woq_engine_time_prefill_gemm",
]

_lib = None


def lib():
    """Load libwoq_hip.so or raise — never falls back to a CPU/eager path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "QBits: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback for the MI355X path." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cs, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float
    L.woq_last_error.restype = ctypes.c_char_p
    L.woq_packed_weight_size.restype = cs
    L.woq_packed_weight_size.argtypes = [ci] * 7
    L.woq_table_digit_planes.restype = ci
    L.woq_table_digit_planes.argtypes = [ci, ci, vp, vp]
    L.woq_repack_quantized_weight.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, cs, vp]
    L.woq_quantize_to_packed_weight.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, cs, vp]
    L.woq_dequantize_packed_weight.argtypes = [vp, ctypes.POINTER(BlobHeader), vp, ci, vp]
    L.woq_read_header.argtypes = [vp, ctypes.POINTER(BlobHeader), vp]
    L.woq_blob_extract.argtypes = [vp, ctypes.POINTER(BlobHeader), ci, vp, vp]
    L.woq_linear.argtypes = [vp, ci, ci, vp, ctypes.POINTER(BlobHeader), vp, vp, ci, ci, ci, vp]
    L.woq_rmsnorm.argtypes = [vp, ci, vp, cf, ci, ci, vp, ci, vp]
    L.woq_rope.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, vp]
    L.woq_silu_mul.argtypes = [vp, vp, ci, cs, vp, vp]
    L.woq_gelu.argtypes = [vp, ci, cs, ci, vp, vp]
    L.woq_engine_create.argtypes = [ctypes.POINTER(EngineConfig), ctypes.POINTER(vp)]
    L.woq_engine_destroy.argtypes = [vp]
    L.woq_engine_destroy.restype = None
    L.woq_engine_set_layer.argtypes = [vp, ci, ctypes.POINTER(LayerWeights)]
    L.woq_engine_set_head.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp]
    for f in ("woq_engine_token_ptr", "woq_engine_pos_ptr", "woq_engine_logits_ptr", "woq_engine_hidden_ptr"):
        getattr(L, f).restype = vp
        getattr(L, f).argtypes = [vp]
    L.woq_engine_bind_io.argtypes = [vp, vp, vp, vp, vp]
    L.woq_engine_step.argtypes = [vp, ci, vp]
    L.woq_engine_capture.argtypes = [vp, ci, vp]
    L.woq_engine_replay.argtypes = [vp, ci, vp]
    L.woq_engine_steps.argtypes = [vp, ci, ci, vp]
    L.woq_engine_set_time_eager.argtypes = [vp, ci]
    L.woq_engine_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.woq_engine_phase.argtypes = [vp, ci, ci, ci, vp]
    L.woq_engine_prefill.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    L.woq_engine_prefill_logits_ptr.restype = vp
    L.woq_engine_prefill_logits_ptr.argtypes = [vp]
    L.woq_engine_set_attn_splits.argtypes = [vp, ci]
    L.woq_engine_attn_splits.argtypes = [vp]
    L.woq_engine_set_attn_grouped.argtypes = [vp, ci]
    L.woq_engine_attn_grouped.argtypes = [vp]
    L.woq_engine_set_attn_chunk.argtypes = [vp, ci]
    L.woq_engine_attn_chunk.argtypes = [vp]
    L.woq_engine_set_fuse_attn.argtypes = [vp, ci]
    L.woq_engine_fuse_attn.argtypes = [vp]
    L.woq_engine_status.argtypes = [vp, vp]
    L.woq_engine_clear_status.argtypes = [vp, vp]
    L.woq_engine_kv_cache_ptr.restype = vp
    L.woq_engine_kv_cache_ptr.argtypes = [vp, ci]
    L.woq_engine_time_gemv.argtypes = [vp, ci, vp, ctypes.POINTER(cf), ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ci)]
    L.woq_engine_time_gemv_mask.argtypes = [vp, ci, ci, vp, ctypes.POINTER(cf), ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ci)]
    L.woq_engine_time_twin.argtypes = [vp, ci, ci, vp, ctypes.POINTER(cf)]
    L.woq_engine_set_tp_options.argtypes = [vp, ci, ci]
    L.woq_engine_time_prefill_gemm.argtypes = [vp, ci, ci, ci, vp, ctypes.POINTER(cf), ctypes.POINTER(cf)]
    L.woq_comm_create.argtypes = [ci, ci, cs, ctypes.POINTER(vp)]
    L.woq_comm_handle.argtypes = [vp, vp, cs]
    L.woq_comm_connect.argtypes = [vp, vp, ctypes.POINTER(ci)]
    L.woq_comm_allreduce_f32.argtypes = [vp, vp, cs, vp]
    L.woq_comm_status.argtypes = [vp, vp, ctypes.POINTER(ci)]
    L.woq_comm_set_timeout_ms.argtypes = [vp, ci]
    L.woq_comm_destroy.argtypes = [vp]
    L.woq_comm_destroy.restype = None
    L.woq_engine_set_comm.argtypes = [vp, vp, ci]
    L.woq_set_workspace.argtypes = [vp, cs]
    L.woq_engine_uses_xq.argtypes = [vp]
    L.woq_engine_token_log_ptr.restype = vp
    L.woq_engine_token_log_ptr.argtypes = [vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().woq_last_error().decode())


def require_gpu():
    """The product path needs a real device: fail loudly instead of silently doing something else."""
    import torch

    if not torch.cuda.is_available() or lib().woq_device_count() == 0:
        raise RuntimeError("QBits: no MI355X/HIP device visible; the gfx950 WOQ path has no CPU fallback")


def torch_dtype_code(dt):
    import torch

    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise RuntimeError("QBits: unsupported qbits data type.")  # text of qbits.cpp:35


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
