"""One prefill GEMM shape, a few launches (for rocprofv3 counter passes). args: M K N group asym(0/1) compute"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402

M, K, N, group, asym = (int(a) for a in sys.argv[1:6])
compute = sys.argv[6] if len(sys.argv) > 6 else "bf16"
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randint(-8, 8, (K, N), generator=g, device="cuda", dtype=torch.int8)
s = (torch.rand(K // group, N, generator=g, device="cuda") + 0.5) * 0.005
z = torch.randint(-8, 8, (K // group, N), generator=g, device="cuda", dtype=torch.int8) if asym else torch.empty(0, dtype=torch.int8)
blob = qbits.repack_quantized_weight(q, s, z, torch.empty(0, dtype=torch.int32), "int4_clip", "fp16", compute, bool(asym), group)
x = torch.randn(M, K, generator=g, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(3):
    qbits.woq_linear(x, blob, torch.empty(0), out, compute, "int4_clip", "fp16", bool(asym))
torch.cuda.synchronize()
