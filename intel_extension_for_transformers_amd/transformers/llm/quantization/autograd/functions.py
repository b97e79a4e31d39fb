"""`matmul_kbit` — the seam between `QuantizedLinearQBits.forward` and the operator module, under the reference's
names (intel_extension_for_transformers/transformers/llm/quantization/autograd/functions.py).

  * `matmul_kbit(A, B, bias, out, compute_dtype, weight_dtype, scale_dtype, scheme, do_dequant=False)` (:184-217):
    inference form = one `qbits.woq_linear` call writing `out` in place;
  * `QBITS_DEBUG` (:108, :203): any value other than unset / "NULL" routes the SAME call through
    `qbits_woq_linear_ref_impl` — dequantise the blob, optional g_idx gather, fp32 matmul, + bias (:41-63) — which is
    the reference's own definition of the result and gives an in-product A/B of the fused kernels at any size.
    Both arms run on the GPU (`dequantize_packed_weight` is a HIP kernel, the matmul is torch's rocBLAS call);
  * `qbits_acquire_type` (:29-38) — selector enum of `acquire_packed_weight_info`.

The training form (`MatMulKBit`, do_dequant=True: an autograd function whose backward dequantises W, :70-181) is out
of scope for this path and raises.
"""
import os
from enum import Enum

import torch

from ..... import qbits


class qbits_acquire_type(Enum):
    SIZE = 0
    BLOCKSIZE = 1
    K = 2
    N = 3
    ACT_SHUFFLE = 4
    G_IDX = 5
    WEI_TYPE = 6
    CMPT_TYPE = 7
    SCALE_TYPE = 8
    SCALE_TENSOR = 9
    ZP_TENSOR = 10
    IS_ASYM = 11


def qbits_woq_linear_ref_impl(activation, packw, bias, compute_type, weight_type, scale_type):
    """functions.py:41-63: W = dequantize_packed_weight(packw) [K, N] fp32; x = index_select(x, 1, g_idx) when the
    blob carries a shuffle; x.float() @ W (+ bias). Returns a new fp32 tensor [M, N]."""
    activation = activation.to(torch.float32)
    n = int(qbits.acquire_packed_weight_info(packw, qbits_acquire_type.N.value)[0])
    k = activation.shape[1]
    revert_wei = torch.empty(k, n, dtype=torch.float32, device=packw.device)
    qbits.dequantize_packed_weight(packw, revert_wei, False, compute_type, weight_type, "fp32")
    if int(qbits.acquire_packed_weight_info(packw, qbits_acquire_type.ACT_SHUFFLE.value)[0]) != 0:
        g_idx = qbits.acquire_packed_weight_info(packw, qbits_acquire_type.G_IDX.value)
        activation = torch.index_select(activation, 1, g_idx.to(activation.device, torch.int64))
    out = torch.matmul(activation, revert_wei)
    if bias is not None and bias.numel() != 0:
        assert bias.is_contiguous()
        assert bias.dtype == torch.float32
        out += bias
    return out


def qbits_debug_enabled():
    """The reference reads the variable on every call (functions.py:108,203), so flipping it mid-run works."""
    return os.getenv("QBITS_DEBUG", "NULL") != "NULL"


def matmul_kbit(A, B, bias, out, compute_dtype, weight_dtype, scale_dtype, scheme, do_dequant=False):
    """functions.py:184-217. Returns `out` (written in place on the fused arm; the reference arm returns its own
    fp32 result cast to `out`'s dtype so that callers see one dtype either way)."""
    if do_dequant:
        raise NotImplementedError("QBits: the training form of matmul_kbit (MatMulKBit, do_dequant=True) is outside "
                                  "the MI355X inference path")
    packw = B.data if isinstance(B, torch.nn.Parameter) else B
    if not qbits_debug_enabled():
        qbits.woq_linear(A, packw, bias if bias is not None else _EMPTY_F32, out, compute_dtype, weight_dtype,
                         scale_dtype, False if scheme == "sym" else True)
        return out
    ref = qbits_woq_linear_ref_impl(A, packw, bias, compute_dtype, weight_dtype, scale_dtype)
    if out is not None and out.shape == ref.shape:
        out.copy_(ref)
        return out
    return ref


_EMPTY_F32 = torch.empty(0, dtype=torch.float32)
