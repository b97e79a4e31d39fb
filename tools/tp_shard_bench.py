"""BASELINE configs[3] on ONE device: the compute of one rank of Llama-2-70B at TP = 8 (hidden 8192, 8 q heads + 1 kv
head, inter 3584, 80 layers, vocab shard 4000), synthetic int4 g128 weights, batch-1 decode without the collectives
(a one-rank group: partial sums = sums). Gives the per-rank compute time a real 8-GPU run adds its 160 all-reduces
to. args: [layers=80]. Development tool."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    d = dict(hidden=8192, inter=3584, heads=8, kv_heads=1, head_dim=128, vocab=4000)
    eng = WoqDecoderEngine(d["hidden"], d["inter"], d["heads"], d["kv_heads"], d["head_dim"], layers, d["vocab"],
                           max_ctx=512)
    synth_llama_weights(eng, d["hidden"], d["inter"], d["heads"], d["kv_heads"], d["head_dim"], layers, d["vocab"],
                        group=128, sym=True, scale_dtype="fp16")
    eng.prefill(torch.randint(0, d["vocab"], (32,)).tolist())
    tok0, pos0 = eng.token.clone(), eng.pos.clone()
    res = {}
    for mode in ("eager", "graph"):  # eager bursts (the default since round 4) and the replayed graph
        if mode == "graph":
            eng.capture(greedy=True)
        run = eng.run if mode == "eager" else eng.replay_graph
        for _ in range(6):  # conditioning + warm-up
            eng.token.copy_(tok0)
            eng.pos.copy_(pos0)
            run(64)
            torch.cuda.synchronize()
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        n = 64
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / n
    dt = res["eager"]
    params = layers * (d["hidden"] * (d["heads"] + 2 * d["kv_heads"]) * d["head_dim"] + d["heads"] * d["head_dim"] * d["hidden"]
                       + 3 * d["hidden"] * d["inter"])
    wbytes = params // 2 + params // 128 * 2
    print(json.dumps(dict(layers=layers, ms_per_token_compute_only=dt * 1e3, ms_per_token_graph_replay=res["graph"] * 1e3, rank_weight_bytes=wbytes,
                          rank_gbps=wbytes / dt / 1e9, allreduces_per_token=2 * layers)))


if __name__ == "__main__":
    main()
