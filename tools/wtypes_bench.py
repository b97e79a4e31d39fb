"""Decode (M = 1) and prompt (M = 2048) timing of qbits.woq_linear per weight type on the Llama-2-7B qkv shape
(K 4096, N 12288, group 128): what the 'other weight dtypes' rows of SURVEY 8(f) cost next to int4_clip."""
import sys
import time

import torch

sys.path.insert(0, ".")
from intel_extension_for_transformers_amd import qbits  # noqa: E402

K, N = 4096, 12288
w = torch.randn(K, N, device="cuda") * 0.02
e = torch.empty(0)
for wt, st in (("int4_clip", "fp32"), ("int8", "fp32"), ("nf4", "fp32"), ("fp4_e2m1", "fp32"), ("fp8_e4m3", "fp32")):
    blobs = [qbits.quantize_to_packed_weight(w + i * 1e-3, False, 128, "fp32", wt, st, False) for i in range(8)]
    for M in (1, 4, 2048):
        x = torch.randn(M, K, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for b in blobs:
            qbits.woq_linear(x, b, e, out, "fp32", wt, st, False)
        torch.cuda.synchronize()
        reps = 10 if M < 100 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            for b in blobs:
                qbits.woq_linear(x, b, e, out, "fp32", wt, st, False)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / (reps * 8) * 1e6
        by = blobs[0].numel()
        print(f"{wt:10s} M={M:5d}  {us:10.1f} us per call   blob {by / 1e6:6.1f} MB   "
              f"{by / us / 1e3:7.1f} GB/s   {2.0 * M * K * N / us / 1e6:8.1f} TFLOP/s", flush=True)
