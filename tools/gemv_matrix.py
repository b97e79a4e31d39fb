"""Decode-GEMV timing matrix (development tool): one Llama-2-7B projection shape through qbits.woq_linear over the
blob / activation variants the boundary accepts, to catch configurations that fall off the fast path.
args: [K=4096] [N=12288]."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 12288
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randint(-8, 8, (K, N), generator=g, device="cuda", dtype=torch.int8)
    e = torch.empty(0)
    print("%-6s %-5s %-6s %-5s %-3s %9s %9s" % ("group", "asym", "scale", "act", "M", "us", "GB/s"))
    for group in (32, 64, 128, -1):
        G = 1 if group == -1 else K // group
        s = (torch.rand(G, N, generator=g, device="cuda") + 0.5) * 0.005
        for asym in (False, True):
            z = torch.randint(-8, 8, (G, N), generator=g, device="cuda", dtype=torch.int8) if asym else torch.empty(0, dtype=torch.int8)
            for scale in ("fp32", "fp16", "bf16"):
                blob = qbits.repack_quantized_weight(q, s, z, torch.empty(0, dtype=torch.int32), "int4_clip", scale,
                                                     "fp32", asym, group)
                for act, M in (("fp32", 1), ("fp32", 2), ("fp32", 4), ("fp32", 8), ("bf16", 1)):
                    if (scale != "fp16" and (M != 1 or act != "fp32")):
                        continue
                    x = torch.randn(M, K, generator=g, device="cuda").to(torch.float32 if act == "fp32" else torch.bfloat16)
                    out = torch.empty(M, N, device="cuda", dtype=x.dtype)
                    for _ in range(5):
                        qbits.woq_linear(x, blob, e, out, "fp32", "int4_clip", scale, asym)
                    torch.cuda.synchronize()
                    n = 100
                    t0 = time.perf_counter()
                    for _ in range(n):
                        qbits.woq_linear(x, blob, e, out, "fp32", "int4_clip", scale, asym)
                    torch.cuda.synchronize()
                    us = (time.perf_counter() - t0) / n * 1e6
                    print("%-6d %-5d %-6s %-5s %-3d %9.2f %9.1f" % (group, asym, scale, act, M, us, blob.numel() / us / 1e3))


if __name__ == "__main__":
    main()
