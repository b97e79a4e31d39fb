"""Is the ~1 us-per-launch bimodality of the captured decode step (r04e / r04f: 0.76 vs 0.87 ms per 16-layer token, stable
inside an engine instance) a property of the hipGraph instantiation or of the memory behind the engine? One engine
(Mistral-7B shape, 16 layers, fp8 KV at 8192 positions by default; `7b` = Llama-2-7B shape, 32 layers, short context):
capture -> time 3 x 64 replays, RE-capture (new hipGraphExec over the same memory) -> time, N times; then the same
timing through EAGER steps (no graph).   python tools/graph_mode_probe.py [mistral|7b] [recaptures=10]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "mistral"
    n_cap = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    vocab = 32000
    if which == "7b":
        hidden, inter, heads, kvh, hd, layers, ctx, kvd = 4096, 11008, 32, 32, 128, 32, 64, torch.float16
    else:
        hidden, inter, heads, kvh, hd, layers, ctx, kvd = 4096, 14336, 32, 8, 128, 16, 8192, torch.float8_e4m3fn
    eng = WoqDecoderEngine(hidden, inter, heads, kvh, hd, layers, vocab, max_ctx=ctx + 512, kv_dtype=kvd)
    synth_llama_weights(eng, hidden, inter, heads, kvh, hd, layers, vocab, group=128, sym=True, scale_dtype="fp16")
    g = torch.Generator().manual_seed(1)
    toks = torch.randint(0, vocab, (ctx,), generator=g).cuda()
    for s0 in range(0, ctx, 2048):
        eng.prefill(toks[s0:s0 + 2048], start_pos=s0, greedy=True)
    eng.tune_attn_for(ctx + 128)
    tok0, pos0 = eng.token.clone(), eng.pos.clone()

    def timed(fn, n=64):
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / n * 1e3, 4)

    for i in range(n_cap):
        eng.capture(greedy=True)
        for _ in range(6):  # conditioning
            timed(eng.replay)
        print(json.dumps({"capture": i, "ms_per_token": [timed(eng.replay) for _ in range(3)]}), flush=True)

    def eager(n):
        for _ in range(n):
            eng.step(greedy=True)

    print(json.dumps({"eager": [timed(eager) for _ in range(3)]}), flush=True)


if __name__ == "__main__":
    main()
