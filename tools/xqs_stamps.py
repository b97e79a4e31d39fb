"""Stage stamps of the product decode GEMV (csrc/woq_gemv_xqs.h, WOQ_XQS_STAMP 0..6) inside the engine's own launches.
Needs a library whose woq_gemv_xq.hip was compiled with -DWOQ_XQS_STAMPS (tools/visits/r06c.sh builds it):
    WOQ_HIP_LIB=tools/lib_xq_stamps.so python tools/xqs_stamps.py
For each projection: one pass of that projection over all 32 layers of a Llama-2-7B-shaped engine, back to back (the
engine's woq_engine_time_gemv_mask); the stamps left in the buffer are those of the LAST launch. Per stage: the median /
p10 / p90 over waves of (stamp - earliest entry stamp of the launch), 100 MHz wall clock."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from intel_extension_for_transformers_amd import _lib as L  # noqa: E402

NAMES = ["entry", "requests out", "parked in LDS", "tile 0 done", "tiles done", "behind barrier", "end"]


def main():
    eng = bench.build_engine(bench.LLAMA2_7B, max_ctx=512)
    bench.feed_prompt(eng, bench.LLAMA2_7B["vocab"], 8)
    lib = L.lib()
    buf = torch.zeros(2048 * 16 * 16, dtype=torch.int64, device="cuda")
    for bit, name in enumerate(("qkv", "o", "gate_up", "down")):
        buf.zero_()
        torch.cuda.synchronize()
        assert lib.woq_xqs_set_probe(ctypes.c_void_p(buf.data_ptr())) == 0
        ms, by, n = eng.time_gemv(reps=1, mask=1 << bit)
        torch.cuda.synchronize()
        assert lib.woq_xqs_set_probe(None) == 0
        st = buf.cpu().numpy().reshape(2048, 16, 16)
        live = st[:, :, 0] > 0
        t0 = st[:, :, 0][live].min()
        print("%s: %d workgroups x %d waves, %.2f us per launch with the stamps on (%.1f MB per launch)"
              % (name, int(live.any(axis=1).sum()), int(live.sum(axis=1).max()), ms * 1e3 / n, by / n / 1e6))
        for k, nm in enumerate(NAMES):
            v = st[:, :, k][live & (st[:, :, k] > 0)]
            if v.size == 0:
                continue
            us = (v - t0) * 0.01
            print("   %-15s min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f   (%d waves)"
                  % (nm, us.min(), np.percentile(us, 10), np.median(us), np.percentile(us, 90), us.max(), v.size))


if __name__ == "__main__":
    main()
