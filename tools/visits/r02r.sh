#!/bin/bash
# r02r: native 16-bit activation rows in the tile GEMV — parity tests + per-call time of qbits.woq_linear
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02r_tests.txt
timeout 300 python - > gpurun_out/r02r_time.txt 2>&1 <<'PY'
import torch, numpy as np
from intel_extension_for_transformers_amd import qbits
K, N = 4096, 11008
w = torch.randn(K, N, device="cuda") * 0.02
blob = qbits.quantize_to_packed_weight(w, False, 128, "fp32", "int4_clip", "fp32", False)
for dt in (torch.float32, torch.bfloat16, torch.float16):
    for M in (1, 4):
        x = torch.randn(M, K, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        e = torch.empty(0)
        for _ in range(20):
            qbits.woq_linear(x, blob, e, out, "fp32", "int4_clip", "fp32", False)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(50):
                    qbits.woq_linear(x, blob, e, out, "fp32", "int4_clip", "fp32", False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        a.record()
        for _ in range(10): g.replay()
        b.record(); torch.cuda.synchronize()
        print(dt, "M", M, "us/call %.2f" % (a.elapsed_time(b) * 1000 / 500))
PY
cat gpurun_out/r02r_tests.txt gpurun_out/r02r_time.txt
