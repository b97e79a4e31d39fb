"""r05d: a GPTQ act-order (g_idx on every projection) Llama-2-7B-shaped engine against the plain int4 one, same box:
tokens/s (three 64-step regions each), the engine's kernel path, per-projection us of the tile GEMV with and without the
gather (woq_linear at M = 1 through the C ABI, the generic fp32 kernel beside it via WOQ_SHUFFLE_GENERIC in a second run)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

cfg = bench.LLAMA2_7B
rows = []
for name, kw in (("int4 g128 (XQ engine)", {}), ("int4 g128 act-order", dict(act_order=True))):
    if os.environ.get("WOQ_ENGINE_XQ") == "0" and not kw:
        name = "int4 g128 (fp32-activation engine, WOQ_ENGINE_XQ=0)"
    eng = bench.build_engine(cfg, max_ctx=512, **kw)
    bench.feed_prompt(eng, cfg["vocab"], 32)
    eng.capture(greedy=True)
    eng.replay_graph(16)
    torch.cuda.synchronize()
    tps = []
    tok0, pos0 = eng.token.clone(), eng.pos.clone()
    for _ in range(3):
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.replay_graph(64)
        torch.cuda.synchronize()
        tps.append(64 / (time.perf_counter() - t0))
    ms, by, n = eng.time_gemv(reps=8)
    print(json.dumps({"engine": name, "tokens_per_s": [round(x, 1) for x in tps], "xq": eng.uses_xq(),
                      "fused_attn": eng.uses_fused_attn(), "status": eng.status(),
                      "gemv_us_per_launch": round(ms * 1e3 / (8 * n), 3),
                      "shuffle_generic": bool(os.environ.get("WOQ_SHUFFLE_GENERIC"))}), flush=True)
    del eng
    bench.free_gpu()
