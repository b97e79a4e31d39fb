"""r05h: one-step graph replays vs k-step unrolled graphs vs eager bursts, same engine, same prompt state; tokens compared."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
cfg = bench.LLAMA2_7B
eng = bench.build_engine(cfg, max_ctx=512)
def run(mode, steps, warm=5, reps=5):
    out = []
    toks = None
    for _ in range(reps):
        bench.feed_prompt(eng, cfg["vocab"], 32)
        if mode != "eager":
            eng.capture(greedy=True)
        f = eng.run if mode == "eager" else eng.replay_graph
        f(warm); torch.cuda.synchronize()
        t0 = time.perf_counter(); f(steps); torch.cuda.synchronize()
        out.append(steps / (time.perf_counter() - t0))
        toks = eng.token_log()[32:32 + warm + steps].cpu().tolist()
    return sorted(out)[len(out) // 2], toks
for steps in (20, 128):
    res = {}
    for unroll in ("1", "4", "8", "16"):
        os.environ["WOQ_ENGINE_GRAPH_UNROLL"] = unroll
        res["graph_unroll_" + unroll], t = run("graph", steps)
        if unroll == "1": ref = t
        assert t == ref, "tokens differ at unroll " + unroll
    res["eager"], t = run("eager", steps)
    assert t == ref
    print(json.dumps({"steps": steps, **{k: round(v, 1) for k, v in res.items()}, "tokens_identical": True}), flush=True)
