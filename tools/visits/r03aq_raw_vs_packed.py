"""GEMM kernel time of the raw-A form (fp16 rows fetched row-major) against the packed-tile form (fp32 rows through the
pack pass) for the o / down / qkv / gate-up shapes at M = 8192 — run under rocprofv3 --kernel-trace --stats."""
import sys
import torch
sys.path.insert(0, ".")
from intel_extension_for_transformers_amd import qbits
M = 8192
e8, e32, e = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32), torch.empty(0)
for K, N in ((4096, 4096), (11008, 4096), (4096, 12288), (4096, 22016)):
    q = torch.randint(-8, 8, (K, N), dtype=torch.int8, device="cuda")
    s = torch.rand(K // 128, N, device="cuda") * 0.01
    blob = qbits.repack_quantized_weight(q, s, e8, e32, "int4_clip", "fp16", "bf16", False, 128)
    for dt in (torch.float16, torch.float32):
        x = torch.randn(M, K, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        for _ in range(6):
            qbits.woq_linear(x, blob, e, out, "bf16", "int4_clip", "fp16", False)
        torch.cuda.synchronize()
