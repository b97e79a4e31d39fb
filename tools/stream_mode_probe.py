"""Is the ~1 us-per-kernel-boundary gap between replayed graphs and eager bursts (profiles/r04g_*) about the GRAPH or about
the STREAM it runs on? runtime/engine.py replays its graph on a stream torch created (capture is not allowed on the legacy
null stream), while eager bursts go to the caller's current stream — the null stream in bench.py. Four combinations on one
Llama-2-7B-shaped engine, 3 x 64 tokens each, ms per token:
   eager / null stream, eager / created stream, graph / null stream, graph / created stream."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intel_extension_for_transformers_amd import _lib as L  # noqa: E402
from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def main():
    hidden, inter, heads, kvh, hd, layers, vocab = 4096, 11008, 32, 32, 128, 32, 32000
    eng = WoqDecoderEngine(hidden, inter, heads, kvh, hd, layers, vocab, max_ctx=512)
    synth_llama_weights(eng, hidden, inter, heads, kvh, hd, layers, vocab, group=128, sym=True, scale_dtype="fp16")
    eng.prefill(torch.randint(0, vocab, (32,)).tolist(), greedy=True)
    eng.capture(greedy=True)
    tok0, pos0 = eng.token.clone(), eng.pos.clone()
    lib, h = L.lib(), eng._h
    created = torch.cuda.Stream()
    prio = torch.cuda.Stream(priority=-1)

    def eager(stream_obj):
        def f(n):
            if stream_obj is None:
                L.check(lib.woq_engine_steps(h, n, 1, None))
            else:
                with torch.cuda.stream(stream_obj):
                    L.check(lib.woq_engine_steps(h, n, 1, ctypes.c_void_p(stream_obj.cuda_stream)))
        return f

    def graph(stream_obj):
        def f(n):
            L.check(lib.woq_engine_replay(h, n, None if stream_obj is None else ctypes.c_void_p(stream_obj.cuda_stream)))
        return f

    def timed(fn, n=64):
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / n * 1e3, 4)

    def via_engine(n):
        eng.replay_graph(n)

    def on_capture_stream_with_waits(n):  # what engine.replay_graph does, spelled out
        cur = torch.cuda.current_stream()
        eng._stream.wait_stream(cur)
        L.check(lib.woq_engine_replay(h, n, ctypes.c_void_p(eng._stream.cuda_stream)))
        cur.wait_stream(eng._stream)

    for name, fn in (("graph / the capture stream, direct C call", graph(eng._stream)),
                     ("graph / capture stream + wait_stream both ways", on_capture_stream_with_waits),
                     ("graph / engine.replay_graph", via_engine),
                     ("eager / null stream", eager(None)), ("eager / created stream", eager(created)),
                     ("eager / high-priority stream", eager(prio)), ("graph / null stream", graph(None)),
                     ("graph / created stream", graph(created)), ("graph / high-priority stream", graph(prio)),
                     ("eager / null stream (again)", eager(None))):
        for _ in range(5):
            timed(fn)
        print(json.dumps({"mode": name, "ms_per_token": [timed(fn) for _ in range(3)]}), flush=True)


if __name__ == "__main__":
    main()
