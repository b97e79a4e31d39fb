"""Captured decode step on synthetic Llama-2-7B-shaped weights: separate launches vs the persistent launch
(csrc/woq_persist.hip). usage: persist_time.py [layers ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from intel_extension_for_transformers_amd.runtime import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def step_us(eng, n=128):
    eng.capture(True)
    eng.replay(8)
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.replay(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    for layers in [int(a) for a in sys.argv[1:]] or [8, 32]:
        eng = WoqDecoderEngine(4096, 11008, 32, 32, 128, layers, 32000, max_ctx=512)
        synth_llama_weights(eng, 4096, 11008, 32, 32, 128, layers, 32000, group=128, sym=True, scale_dtype="fp16")
        res = {}
        for mode in ("launches", "persist"):
            eng.set_persist(mode == "persist")
            if mode == "persist" and not eng.uses_persist():
                from intel_extension_for_transformers_amd import _lib as L
                print("persist not used:", L.lib().woq_last_error().decode())
                break
            eng.reset(1, 0)
            logs = []
            for _ in range(12):
                eng.step(True)
            logs.append(eng.logits.clone())
            us = step_us(eng)
            eng.reset(1, 0)
            eng.replay(40)
            torch.cuda.synchronize()
            logs.append(eng.logits.clone())
            res[mode] = (torch.stack(logs), eng.token_log()[:41].clone(), us, eng.status())
            print(f"layers {layers:2d} {mode:9s}: {us:8.1f} us/step  {1e6 / us:7.1f} tok/s  status {res[mode][3]}", flush=True)
        if len(res) == 2:
            a, b = res["launches"], res["persist"]
            err = ((a[0] - b[0]).abs().amax(dim=1) / a[0].abs().amax(dim=1)).tolist()
            same = (a[1] == b[1]).int().tolist()
            first = same.index(0) if 0 in same else -1
            print(f"   rel logit diff after 12 eager / 40 replayed steps: {err}; first differing token: {first}")
        del eng
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
