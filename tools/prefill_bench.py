"""Prefill-side measurement (BASELINE configs[2]): int4 x fp MFMA GEMM, M tokens through one Llama-2-7B layer's four
fused linears. Prints achieved TFLOP/s per shape against the 2.5 PFLOP/s dense fp16 MFMA peak (MI355X_MICROARCH.md).
Not the headline bench (that is bench.py); a development / profile tool."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    # args: M [group [computes [asym 0/1]]] — a group without the 4th argument means asymmetric (configs[2])
    group, asym = (int(sys.argv[2]), True) if len(sys.argv) > 2 else (128, False)
    if len(sys.argv) > 4:
        asym = bool(int(sys.argv[4]))
    res = []
    computes = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("bf16", "fp32")
    for compute in computes:
        for name, K, N in (("qkv", 4096, 12288), ("o", 4096, 4096), ("gate_up", 4096, 22016), ("down", 11008, 4096)):
            g = torch.Generator(device="cuda").manual_seed(0)
            q = torch.randint(-8, 8, (K, N), generator=g, device="cuda", dtype=torch.int8)
            s = (torch.rand(K // group, N, generator=g, device="cuda") + 0.5) * 0.005
            z = torch.randint(-8, 8, (K // group, N), generator=g, device="cuda", dtype=torch.int8) if asym else \
                torch.empty(0, dtype=torch.int8)
            blob = qbits.repack_quantized_weight(q, s, z, torch.empty(0, dtype=torch.int32), "int4_clip", "fp16", compute,
                                                 asym, group)
            x = torch.randn(M, K, generator=g, device="cuda")
            out = torch.empty(M, N, device="cuda")
            e = torch.empty(0)
            for _ in range(2):
                qbits.woq_linear(x, blob, e, out, compute, "int4_clip", "fp16", asym)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = 5
            for _ in range(n):
                qbits.woq_linear(x, blob, e, out, compute, "int4_clip", "fp16", asym)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            tf = 2.0 * M * K * N / dt / 1e12
            res.append(dict(shape=name, M=M, K=K, N=N, compute=compute, group=group, asym=asym, ms=dt * 1e3,
                            tflops=tf, frac_of_2500=tf / 2500.0))
            print("%-8s M=%d K=%d N=%d compute=%s g%d %s: %8.3f ms  %7.1f TFLOP/s (%.1f%% of 2.5 PF dense)" %
                  (name, M, K, N, compute, group, "asym" if asym else "sym", dt * 1e3, tf, 100 * tf / 2500))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
