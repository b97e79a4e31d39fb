"""Developer micro-benchmark (not the driver's bench.py): per-shape GEMV timings + a decode step on synthetic
Llama-2-7B-shaped weights. Prints achieved algorithmic GB/s."""
import sys
import time

import torch

sys.path.insert(0, ".")
from intel_extension_for_transformers_amd import qbits  # noqa: E402
from intel_extension_for_transformers_amd.runtime import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def bench_shape(K, N, group=128, n_mats=24, iters=20):
    e8, e32 = torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32)
    blobs = []
    for i in range(n_mats):
        q = torch.randint(-8, 8, (K, N), dtype=torch.int8, device="cuda")
        s = torch.rand(K // group, N, device="cuda") * 0.01
        blobs.append(qbits.repack_quantized_weight(q, s, e8, e32, "int4_clip", "fp16", "fp32", False, group))
    x = torch.randn(1, K, device="cuda")
    out = torch.zeros(1, N, device="cuda")
    for b in blobs:
        qbits.woq_linear(x, b, e8, out, "fp32", "int4_clip", "fp16", False)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        for b in blobs:
            qbits.woq_linear(x, b, e8, out, "fp32", "int4_clip", "fp16", False)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / (iters * n_mats)
    by = K * N / 2 + (K // group) * N * 2
    print(f"gemv K={K} N={N}: {us:7.2f} us/launch (back-to-back, incl. gaps)  {by / us / 1e3:7.1f} GB/s algorithmic")


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    import os
    if not os.environ.get("SKIP_SHAPES"):
        for K, N in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
            bench_shape(K, N)
    eng = WoqDecoderEngine(4096, 11008, 32, 32, 128, layers, 32000, max_ctx=512)
    synth_llama_weights(eng, 4096, 11008, 32, 32, 128, layers, 32000, group=128, sym=True, scale_dtype="fp16")
    eng.reset(1, 0)
    for _ in range(3):
        eng.step(True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 32
    for _ in range(n):
        eng.step(True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"eager step: {dt * 1e3:.3f} ms/token  ({1 / dt:.1f} tok/s)")
    with torch.cuda.stream(eng._stream):
        for _ in range(3):
            eng.step(True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            eng.step(True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
    print(f"eager step on side stream: {dt * 1e3:.3f} ms/token  ({1 / dt:.1f} tok/s)")
    eng.capture(True)
    eng.replay(3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 64
    eng.replay(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    wbytes = layers * (4096 * 12288 + 4096 * 4096 + 4096 * 22016 + 11008 * 4096) * (0.5 + 2 / 128)
    print(f"graph step: {dt * 1e3:.3f} ms/token  ({1 / dt:.1f} tok/s)  quantized-weight stream {wbytes / dt / 1e9:.0f} GB/s")
    ms, by, nl = eng.time_gemv(3)
    print(f"gemv-only (event pairs): {ms / 3:.3f} ms per pass over {nl} launches, {by / (ms / 3) / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
