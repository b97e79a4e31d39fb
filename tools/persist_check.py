"""Persistent token kernel (csrc/woq_persist.hip) against the separate launches on a Llama-2-7B-shaped decoder:
logits / greedy tokens after a prompt pass, over eager steps and graph replays, and the captured step's time."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from test_gpu_fullsize_oracle import build_7b_shape  # noqa: E402


def run(layers, group, asym, kv_dtype, prompt_len=40, steps=4, replays=24):
    eng, _, cfg = build_7b_shape(layers, group, asym, kv_dtype=kv_dtype, max_ctx=512)
    rng = np.random.default_rng(5)
    prompt = rng.integers(0, cfg["vocab"], prompt_len).tolist()
    out = {}
    for mode in ("launches", "persist", "persist again"):
        eng.set_persist(mode != "launches")
        used = eng.uses_persist()
        if mode != "launches" and not used:
            from intel_extension_for_transformers_amd import _lib as L
            print("persist not used:", L.lib().woq_last_error().decode())
            return
        eng.prefill(prompt, greedy=True)
        logs = []
        for _ in range(steps):
            eng.step(greedy=True)
            logs.append(eng.logits.clone())
        torch.cuda.synchronize()
        st0 = eng.status()
        eng.capture(greedy=True)
        eng.replay(replays)
        torch.cuda.synchronize()
        logs.append(eng.logits.clone())
        toks = eng.token_log()[prompt_len:prompt_len + steps + replays + 1].clone()
        # time: replays back to back
        eng.prefill(prompt, greedy=True)
        eng.replay(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.replay(64)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 64 * 1e6
        out[mode] = (torch.stack(logs), toks, us)
        print(f"  {mode:14s} status {st0}->{eng.status()}  {us:8.1f} us/step", flush=True)
    ref, got, again = out["launches"], out["persist"], out["persist again"]
    err = (got[0] - ref[0]).abs().max().item() / ref[0].abs().max().item()
    print(f"  max |dlogit| / max |logit| = {err:.2e}; tokens equal: {torch.equal(got[1], ref[1])}; "
          f"persist repeatable: {torch.equal(got[0], again[0])}")
    print(f"  per-layer: launches {(ref[2]) / layers:.2f} us, persist {(got[2]) / layers:.2f} us (incl. head / embed share)")


if __name__ == "__main__":
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    cases = [(128, False, torch.float16)]
    if which == "all":
        cases += [(32, True, torch.float16), (128, False, torch.float8_e4m3fn)]
    for group, asym, kv in cases:
        print(f"layers {layers} group {group} asym {asym} kv {kv}", flush=True)
        run(layers, group, asym, kv)
