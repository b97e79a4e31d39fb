"""M = 5..8 rows: two launches of the tile GEMV (weights streamed twice) against the MFMA GEMM (streamed once).
Run twice: default and WOQ_GEMV_AS_GEMM=1. Development tool."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402

shapes = [("qkv", 4096, 12288), ("o", 4096, 4096), ("gate_up", 4096, 22016), ("down", 11008, 4096)]
e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
for name, K, N in shapes:
    blobs = []
    for i in range(8):
        q = torch.randint(-8, 8, (K, N), dtype=torch.int8, device="cuda")
        s = torch.rand(K // 128, N, device="cuda") * 0.01
        blobs.append(qbits.repack_quantized_weight(q, s, e8, e32, "int4_clip", "fp16", "bf16", False, 128))
    for M in (4, 5, 8, 9, 12, 16, 17):
        x = torch.randn(M, K, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for b in blobs:
            qbits.woq_linear(x, b, torch.empty(0), out, "bf16", "int4_clip", "fp16", False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            for b in blobs:
                qbits.woq_linear(x, b, torch.empty(0), out, "bf16", "int4_clip", "fp16", False)
        torch.cuda.synchronize()
        print("%-8s M=%2d  %7.2f us per call  (WOQ_GEMV_AS_GEMM=%s)" % (
            name, M, (time.perf_counter() - t0) / 80 * 1e6, os.environ.get("WOQ_GEMV_AS_GEMM", "0")))
