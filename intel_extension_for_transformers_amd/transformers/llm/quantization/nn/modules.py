"""`QuantizedLinearQBits` for MI355X — the module `_replace_linear` swaps in for every `nn.Linear`.

Same class name, constructor, methods and attributes as the reference's CPU module
(intel_extension_for_transformers/transformers/llm/quantization/nn/modules.py:92-392), so code written against the
reference (`isinstance(m, QuantizedLinearQBits)`, `m.set_weights_bias(...)`, `m.recover_qparms()`) keeps working;
underneath, `forward` is one `qbits.woq_linear` call into libwoq_hip.so on the tensor's HIP stream.

Differences that are deliberate (SURVEY.md §3.2 "overheads the GPU build should not inherit"):
  * the output buffer is `torch.empty`, not `torch.zeros` (modules.py:146): the kernel writes every element;
  * the packed blob's header is parsed once per tensor, not per call (bestla_weightonly_dispatcher.cpp:335);
  * there is no CPU fallback: without the HIP library / a GPU the module raises (parity claims depend on it).
"""
import os

import torch

from ..... import qbits
from ..... import _lib as L
from ..autograd import matmul_kbit


class ParamsQBits(torch.nn.Parameter):
    """nn.Parameter subclass carrying the opaque packed blob (reference modules.py:67-89): `model.to()` /
    `state_dict()` see an ordinary int8 tensor; the quantisation attributes ride along as Python attributes."""

    def __new__(cls, data=None, requires_grad=False, quant_state=None, blocksize=32, compress_statistics=True,
                quant_dtype=None, scale_dtype="fp32"):
        if data is None:
            data = torch.empty(0)
        self = torch.Tensor._make_subclass(cls, data, requires_grad)
        self.blocksize = blocksize
        self.compress_statistics = compress_statistics
        self.quant_dtype = quant_dtype
        self.scale_dtype = scale_dtype
        self.quant_state = quant_state
        return self


def _cfg(q_config, name, default=None):
    return getattr(q_config, name, default) if q_config is not None else default


class QuantizedLinearQBits(torch.nn.Linear):
    """int4 weight-only linear: y = x @ dequant(W) + b with W a WQH1 blob in HBM."""

    def __init__(self, input_features, output_features, bias=True, compute_dtype="fp32", compress_statistics=True,
                 weight_dtype="int4_clip", bits=4, scale_dtype="fp32", blocksize=32, scheme="sym", device=None,
                 double_quant_scale_dtype=None, compression_dtype=torch.int32, compression_dim=1,
                 use_optimum_format=False):
        # the dense weight of nn.Linear is never materialised: it is replaced by the packed blob below
        super().__init__(input_features, output_features, bias, device="meta")
        self.weight = ParamsQBits(torch.empty(0, dtype=torch.int8), requires_grad=False)
        self.bias = None
        self._wants_bias = bool(bias)
        self.device = device
        self.compute_dtype = compute_dtype
        self.compress_statistics = compress_statistics
        self.blocksize = blocksize
        self.scheme = scheme
        self.weight_dtype = weight_dtype
        self.bits = bits
        self.scale_dtype = scale_dtype
        self.double_quant_scale_dtype = double_quant_scale_dtype
        self.compression_dim = compression_dim
        if compression_dtype not in (torch.int8, torch.int16, torch.int32, torch.int64):
            raise AssertionError("Only support torch.int8|16|32|64 as compressed dtype.")
        self.use_optimum_format = use_optimum_format
        if use_optimum_format:  # GPTQ-style checkpoints: [K, N] weights, fp16 scales, int32 words
            compression_dtype = torch.int32
        self.compression_dtype = compression_dtype
        self.n_pack = compression_dtype.itemsize * 8 // bits
        self._bias32 = None

    # ---- the hot call (reference modules.py:140-169) --------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if self.weight.numel() == 0:
            raise RuntimeError("QBits: QuantizedLinearQBits has no packed weight (call set_weights_bias first)")
        shape = x.shape[:-1]
        m = 1
        for d in shape:
            m *= int(d)
        x2 = x.reshape(m, x.shape[-1])
        # the reference boundary always holds fp32 activations (modules.py:152-154); here fp16 / bf16 rows go down as
        # they are — the kernels widen them in registers, which is the same exact conversion without the cast launch
        if x2.dtype not in _FLOAT_OUT and x2.dtype != torch.float32:
            x2 = x2.float()
        elif not x2.is_contiguous():
            x2 = x2.contiguous()
        out = torch.empty(m, self.out_features, dtype=x.dtype if x.dtype in _FLOAT_OUT else torch.float32,
                          device=x.device)
        bias = self._bias32
        if bias is None and self.bias is not None:
            bias = self._bias32 = self.bias.detach().to(x.device, torch.float32).contiguous()
        # reference modules.py:155-164: matmul_kbit -> qbits.woq_linear, or the dequantise -> matmul reference arm
        # under QBITS_DEBUG (autograd/functions.py:184-217)
        out = matmul_kbit(x2, self.weight, bias, out, self.compute_dtype, self.weight_dtype, self.scale_dtype,
                          self.scheme, do_dequant=False)
        out = out.view(*shape, self.out_features)
        if os.environ.get("backend", None) == "use_vllm":  # vLLM's linear layers return (output, output_bias):
            return out, None                               # the reference's seam, modules.py:166-167
        return out

    # ---- load-time (reference modules.py:171-262) -----------------------------------------------------------------
    def _adopt(self, packw, bias):
        self.weight = ParamsQBits(data=packw, requires_grad=False, quant_state={"scheme": self.scheme},
                                  blocksize=self.blocksize, compress_statistics=self.compress_statistics,
                                  quant_dtype=self.weight_dtype, scale_dtype=self.scale_dtype)
        self._bias32 = None
        if bias is not None:
            self.bias = torch.nn.Parameter(bias.detach().to(packw.device), requires_grad=False)
        else:
            self.bias = None

    def set_fp_weights_bias(self, weight_data, bias=None):
        """RTN of an fp weight in nn.Linear layout [N, K] straight into a blob on the device
        (reference modules.py:171-193 -> qbits.quantize_to_packed_weight)."""
        packw = qbits.quantize_to_packed_weight(weight_data.detach().float(), True, self.blocksize, self.compute_dtype,
                                                self.weight_dtype, self.scale_dtype, self.scheme == "asym")
        self._adopt(packw, bias)

    def set_weights_bias(self, int_weight, gptq_scales, gptq_zeros, g_idx, q_config, bias=None):
        """int weights [K, N] in the UNSIGNED domain (0..15) + scales [G, N] + zeros [G, N] (unsigned, already +1
        un-biased by unpack_weight) + GPTQ g_idx -> blob. Follows reference modules.py:195-262 step by step:
        desc_act row regrouping (:205-224), the signed-nibble shift (:225-227), sym drops zeros (:233-234),
        non-GPTQ drops g_idx (:236-237), then qbits.repack_quantized_weight (:239-249)."""
        method = _cfg(q_config, "quant_method")
        method = getattr(method, "value", method)
        sym = bool(_cfg(q_config, "sym", self.scheme == "sym"))
        no_idx = torch.empty(0, dtype=torch.int32)
        if method == "gptq" and _cfg(q_config, "desc_act", False) and not _cfg(q_config, "static_groups", False):
            # rows of one group are scattered over K by act-order: gather them so group g owns rows
            # [g*group, (g+1)*group), in order of appearance — then the kernel shuffles activations with g_idx
            gi = g_idx.to(torch.int64).cpu()
            order = torch.argsort(gi, stable=True)
            int_weight = int_weight.to(order.device)[order].contiguous()
        elif method != "gptq" or not _cfg(q_config, "desc_act", False) or _cfg(q_config, "static_groups", False):
            g_idx = no_idx
        if self.bits == 4 and "f" not in self.weight_dtype:
            int_weight = _signed_nibble(int_weight)
            if gptq_zeros is not None and gptq_zeros.numel():
                gptq_zeros = _signed_nibble(gptq_zeros)
        if sym or gptq_zeros is None:
            gptq_zeros = torch.empty(0, dtype=torch.int8)
        packw = qbits.repack_quantized_weight(int_weight.contiguous(), gptq_scales.float().contiguous(),
                                              gptq_zeros.contiguous(), g_idx.contiguous(), self.weight_dtype,
                                              self.scale_dtype, self.compute_dtype, not sym, self.blocksize)
        self._adopt(packw, bias)

    # ---- save path (reference modules.py:297-392) -----------------------------------------------------------------
    def recover_qparms_kn(self):
        """blob -> (int_weight [K, N] int8: unsigned 0..15 for 4 bits, signed for 8; scales fp32 [G, N]; zeros [G, N]
        in the same domain or None; g_idx int32 [K] = the GPTQ group id of every ORIGINAL row, or None) — the tensors
        `set_weights_bias` takes, in its orientation, so that `set_weights_bias(*recover_qparms_kn(), q_config)` is the
        identity (act-order included: the blob keeps rows regrouped and the converted shuffle; both are undone here
        like the reference's recover_idx / recover_int_weight, modules.py:299-326,375-377).
        The reference dequantises and re-quantises with the stored scales (:356-372) because BesTLA cannot hand the
        integers back; the WQH1 blob can, so they are recovered exactly: q = dequant / scale + zp from its own tensors."""
        if self.weight_dtype in ("nf4", "fp4_e2m1", "fp4_e2m1_bnb", "fp8_e4m3", "fp8_e5m2"):
            raise NotImplementedError("QBits: float weight types (%s) have no integer export format" % self.weight_dtype)
        w = self.weight.data
        info = lambda t: qbits.acquire_packed_weight_info(w, t)  # noqa: E731
        k, n = int(info(2)[0]), int(info(3)[0])
        scales = info(9)
        asym = bool(int(info(11)[0]))
        zeros = info(10) if asym else None
        shuffle = info(5).to(torch.int64) if int(info(4)[0]) else None  # position j of the blob = original row shuffle[j]
        deq = torch.empty(k, n, dtype=torch.float32, device=w.device)
        qbits.dequantize_packed_weight(w, deq, False, self.compute_dtype, self.weight_dtype, self.scale_dtype)
        group = int(info(1)[0])
        rows = torch.arange(k, device=w.device) // group
        s = scales[rows]
        safe = torch.where(s == 0, torch.ones_like(s), s)
        q = torch.round(deq / safe)
        if zeros is not None:
            q = q + zeros[rows].float()
        g_idx = None
        if shuffle is not None:
            g_idx = torch.empty(k, dtype=torch.int32, device=w.device)
            g_idx[shuffle] = rows.to(torch.int32)  # recover_idx (:299-305)
            q_orig = torch.empty_like(q)
            q_orig[shuffle] = q                     # recover_int_weight (:307-326): back to the checkpoint's row order
            q = q_orig
        if self.bits == 8:  # the reference un-biases only 4-bit integers (:349-352); int8 stays signed
            return q.clamp_(-128, 127).to(torch.int8), scales, zeros, g_idx
        int_weight = (q + 8).clamp_(0, 15).to(torch.int8)  # back to the unsigned domain (recover_qparms :349-352)
        return int_weight, scales, (zeros + 8 if zeros is not None else None), g_idx

    def recover_qparms(self):
        """The reference's 12-tuple, in its orientation (modules.py:378-392): (group_size, in_features, out_features,
        desc_act, g_idx, weight_dtype, bits, scales_dtype, scales [N, G], zp, qzeros [N, G] | None, int_weight [N, K]).
        Asymmetric 4-bit integers and zero points are unsigned (:349-352 adds 8 to the zero points only, and
        quant_weight_w_scale adds the zero point to round(w / scale)); SYMMETRIC ones have no zero point there, so
        the reference's int_weight is the signed round(w / scale) in [-8, 7]; 8-bit zero points uint8 (:353-354)."""
        int_weight, scales, zeros, g_idx = self.recover_qparms_kn()
        if zeros is None and self.bits == 4:
            int_weight = int_weight - 8  # recover_qparms_kn works in the unsigned domain throughout
        k, n = int_weight.shape
        group = int(qbits.acquire_packed_weight_info(self.weight.data, 1)[0])
        qzeros = None
        if zeros is not None:
            qzeros = zeros if self.bits == 4 else (zeros.to(torch.int32) + 128).to(torch.uint8)
        return (group, k, n, g_idx is not None, g_idx, self.weight_dtype, self.bits,
                torch.float32 if self.scale_dtype == "fp32" else None, scales.t(), zeros is not None,
                qzeros.t() if qzeros is not None else None, int_weight.t())


_FLOAT_OUT = (torch.float32, torch.bfloat16, torch.float16)
_EMPTY_F32 = torch.empty(0, dtype=torch.float32)


def _signed_nibble(t):
    """`(t - 8) * 16 // 16` on int8 (reference modules.py:225-227): 0..15 -> -8..7, and 16 wraps to -8
    (pinned by tests/golden/set_weights_bias.npz)."""
    t8 = t.to(torch.int8)
    return torch.div((t8 - 8) * 16, 16, rounding_mode="floor").to(torch.int8)


# the name a GPU-first user would look for
QuantizedLinearHIP = QuantizedLinearQBits
