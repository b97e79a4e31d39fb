"""One prefill GEMM call timed alone: M x K x N through qbits.woq_linear (compute bf16: the one-product MFMA form),
fp32 rows (pack pass + packed-A kernel) or fp16 rows (raw-A kernel). args: M K N [f32|f16] [group] [asym]. Development
tool for A/B library builds (WOQ_HIP_LIB=...)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402


def main():
    M, K, N = (int(v) for v in sys.argv[1:4])
    adt = torch.float16 if len(sys.argv) > 4 and sys.argv[4] == "f16" else torch.float32
    group = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    asym = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
    rng = np.random.default_rng(0)
    q = torch.from_numpy(rng.integers(-8, 8, (K, N), dtype=np.int8)).cuda()
    s = torch.from_numpy((rng.random((K // group, N), dtype=np.float32) * 0.01 + 0.001)).cuda()
    z = torch.from_numpy(rng.integers(-8, 8, (K // group, N), dtype=np.int8)).cuda() if asym else torch.empty(0, dtype=torch.int8)
    blob = qbits.repack_quantized_weight(q, s, z, torch.empty(0, dtype=torch.int32), "int4_clip", "fp16", "bf16", asym, group)
    x = torch.randn(M, K, device="cuda").to(adt)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ws = torch.empty(qbits_ws(M, K, N), dtype=torch.uint8, device="cuda") if False else None
    for _ in range(3):
        qbits.woq_linear(x, blob, torch.empty(0), out, "bf16", "int4_clip", "fp16", asym)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        qbits.woq_linear(x, blob, torch.empty(0), out, "bf16", "int4_clip", "fp16", asym)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("M %d K %d N %d %s g%d asym %d: %.1f us per call (pack pass included), %.0f TFLOP/s"
          % (M, K, N, "f16" if adt == torch.float16 else "f32", group, asym, us, 2.0 * M * K * N / us / 1e6))


if __name__ == "__main__":
    main()
