"""Prefill-GEMM timing matrix (development tool): one shape through qbits.woq_linear over blob variants and compute
types, to catch configurations that fall off the fast path. args: [M=4096] [K=4096] [N=4096]."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    only = sys.argv[4] if len(sys.argv) > 4 else None  # restrict to one compute type
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randint(-8, 8, (K, N), generator=g, device="cuda", dtype=torch.int8)
    e = torch.empty(0)
    print("%-6s %-5s %-6s %-7s %-5s %9s %9s" % ("group", "asym", "scale", "compute", "act", "ms", "TFLOP/s"))
    for group in (32, 64, 128, -1):
        G = 1 if group == -1 else K // group
        s = (torch.rand(G, N, generator=g, device="cuda") + 0.5) * 0.005
        for asym in (False, True):
            z = torch.randint(-8, 8, (G, N), generator=g, device="cuda", dtype=torch.int8) if asym else torch.empty(0, dtype=torch.int8)
            for scale, compute, act in (("fp16", "bf16", "fp32"), ("fp32", "bf16", "bf16"), ("bf16", "fp16", "fp16"),
                                        ("fp16", "fp32", "fp32"), ("fp16", "int8", "fp32")):
                if only and compute != only:
                    continue
                blob = qbits.repack_quantized_weight(q, s, z, torch.empty(0, dtype=torch.int32), "int4_clip", scale,
                                                     compute, asym, group)
                dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[act]
                x = torch.randn(M, K, generator=g, device="cuda").to(dt)
                out = torch.empty(M, N, device="cuda", dtype=dt)
                for _ in range(5):
                    qbits.woq_linear(x, blob, e, out, compute, "int4_clip", scale, asym)
                torch.cuda.synchronize()
                n = 20
                t0 = time.perf_counter()
                for _ in range(n):
                    qbits.woq_linear(x, blob, e, out, compute, "int4_clip", scale, asym)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / n * 1e3
                print("%-6d %-5d %-6s %-7s %-5s %9.3f %9.1f" % (group, asym, scale, compute, act, ms,
                                                               2.0 * M * K * N / ms / 1e9))


if __name__ == "__main__":
    main()
