"""r05a: (1) decode launches out of the Infinity Cache — each projection of the Llama-2-7B layer cold vs read by another
kernel one launch earlier (woq_engine_mall_probe), GEMV and load-only twin; (2) the token-long prefetcher beside the
step (csrc/woq_prefetch.hip): tokens/s off vs on over a parameter sweep, greedy tokens compared with the baseline's.
Usage: python tools/visits/r05a_mall.py [probe] [sweep] [quick]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

what = set(sys.argv[1:]) or {"probe", "sweep"}
cfg = bench.LLAMA2_7B
eng = bench.build_engine(cfg, max_ctx=512)
PROMPT, WARM, STEPS = 32, 16, 128
names = ["qkv", "o", "gate_up", "down"]
print(json.dumps({"lib": os.environ.get("WOQ_HIP_LIB", "default")}), flush=True)

if "probe" in what:
    bench.feed_prompt(eng, cfg["vocab"], PROMPT)
    for lead in (1, 2):
        for twin in (False, True):
            row = {}
            for j, n in enumerate(names):
                cold, hot, rd = eng.mall_probe(j, twin=twin, lead=lead, reps=10)
                row[n] = {"cold_us": round(cold, 2), "hot_us": round(hot, 2), "reader_us": round(rd, 2)}
            print(json.dumps({"probe": "twin" if twin else "gemv", "lead": lead, **row}), flush=True)


def run(tag, **pf):
    bench.feed_prompt(eng, cfg["vocab"], PROMPT)
    if pf:
        eng.set_prefetch(True, **pf)
    else:
        eng.set_prefetch(False)
    eng.capture(greedy=True)
    eng.replay_graph(WARM)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.replay_graph(STEPS)
        torch.cuda.synchronize()
        best.append(STEPS / (time.perf_counter() - t0))
    toks = eng.token_log()[PROMPT:PROMPT + WARM + STEPS].cpu().tolist()
    return best, toks


if "sweep" in what:
    base, ref = run("off")
    print(json.dumps({"prefetch": "off", "tokens_per_s": [round(x, 1) for x in base], "status": eng.status()}), flush=True)
    grid = []
    quick = "quick" in what
    for waves, depth in ((4, 16), (2, 16), (1, 16), (4, 8), (4, 32), (2, 32)):
        grid.append(dict(waves=waves, depth=depth, lead=2, lead_kind=-1))
    for lead, lk in ((1, 3), (2, 0), (2, 1), (2, 2), (3, -1)):
        grid.append(dict(waves=4, depth=16, lead=lead, lead_kind=lk))
    grid.append(dict(waves=4, depth=16, lead=2, lead_kind=-1, wrap=False))
    grid.append(dict(waves=4, depth=16, lead=2, lead_kind=-1, head_mb=96))
    grid.append(dict(waves=4, depth=16, lead=2, lead_kind=-1, grid=512))
    grid.append(dict(waves=2, depth=16, lead=2, lead_kind=-1, grid=128))
    if quick:
        grid = grid[:3]
    for pf in grid:
        r, toks = run("on", **pf)
        # the first replays after a capture run 3 x WARM + STEPS tokens from the same prompt state: compare the first window
        same = toks == ref
        print(json.dumps({"prefetch": pf, "in_use": eng.uses_prefetch(), "tokens_per_s": [round(x, 1) for x in r],
                          "same_tokens": same, "status": eng.status()}), flush=True)
    again, _ = run("off")
    print(json.dumps({"prefetch": "off (again)", "tokens_per_s": [round(x, 1) for x in again]}), flush=True)
