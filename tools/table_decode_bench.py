"""Decode-row timings of the 4-bit table weight types (nf4 / fp4) against int4 on the Llama-2-7B projections, and the
batch-1 decode rate of a Llama-2-7B-shaped engine with nf4 / fp4 layers (round 4: digit-plane unpack on the int8 MFMA,
csrc/woq_gemv_common.h LutArgs). WOQ_TABLE_GENERIC=1 in the environment selects the fp32 VALU kernel these types ran
on through round 3 (separate process: the switch is read once). `--types fp8_e4m3,fp8_e5m2` times the 8-bit float
weights (csrc/woq_gemv_fp8.hip; WOQ_FP8_GENERIC=1: the lookup kernel); a type may carry the compute type its blobs are
packed for, "nf4:bf16" (two digit planes instead of three).

    python tools/table_decode_bench.py [--engine] [--layers 32] [--types int4_clip,nf4,nf4:bf16,fp4_e2m1,fp8_e4m3]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402
from intel_extension_for_transformers_amd.runtime import WoqDecoderEngine  # noqa: E402
from intel_extension_for_transformers_amd.runtime.engine import synth_llama_weights  # noqa: E402

SHAPES = {"qkv": (4096, 12288), "o": (4096, 4096), "gate_up": (4096, 22016), "down": (11008, 4096)}


def split(spec):  # "nf4:bf16" -> ("nf4", "bf16"): weight type and the compute type recorded in the blobs
    return (spec.split(":") + ["fp32"])[:2]


def time_linear(spec, K, N, M, reps=200, group=128):
    wname, cname = split(spec)
    g = torch.Generator(device="cuda").manual_seed(1)
    table = wname != "int4_clip"
    if wname.startswith("fp8"):  # code bytes: positive finite codes of either grid
        q = torch.randint(0, 0x78, (K, N), generator=g, device="cuda", dtype=torch.int8)
    else:
        q = torch.randint(0 if table else -8, 16 if table else 8, (K, N), generator=g, device="cuda", dtype=torch.int8)
    s = (0.5 + torch.rand(K // group, N, generator=g, device="cuda")) * 0.005
    blob = qbits.repack_quantized_weight(q, s, torch.empty(0, dtype=torch.int8), torch.empty(0, dtype=torch.int32),
                                         wname, "fp32" if wname.startswith("fp8") else "fp16", cname, False, group)
    x = torch.randn(M, K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    e = torch.empty(0)
    for _ in range(20):
        qbits.woq_linear(x, blob, e, out, "fp32", wname, "fp16", False)  # (the type strings are read from the blob)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        qbits.woq_linear(x, blob, e, out, "fp32", wname, "fp16", False)
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) * 1e3 / reps


def engine_rate(spec, layers, steps=64):
    wname, cname = split(spec)
    eng = WoqDecoderEngine(4096, 11008, 32, 32, 128, layers, 32000, max_ctx=512)
    synth_llama_weights(eng, 4096, 11008, 32, 32, 128, layers, 32000, group=128, sym=True, weight_dtype=wname,
                        compute_dtype=cname)
    eng.reset(token=1, pos=0)
    eng.run(48, greedy=True)
    eng.capture(greedy=True)
    eng.replay_graph(16)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        eng.reset(token=1, pos=48)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        eng.replay_graph(steps)
        t1.record()
        t1.synchronize()
        best = min(best, t0.elapsed_time(t1) / steps)
    return best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", action="store_true")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--group", type=int, default=128, help="group size of the per-projection timings (fp8: 128 / 64 / 32)")
    ap.add_argument("--types", default="int4_clip,nf4,nf4:bf16,fp4_e2m1,fp4_e2m1_bnb")
    args = ap.parse_args()
    mode = "generic fp32 VALU kernel" if os.environ.get("WOQ_TABLE_GENERIC") else "digit-plane MFMA kernel"
    if os.environ.get("WOQ_FP8_GENERIC"):
        mode += "; fp8: lookup kernel"
    for wname in args.types.split(","):
        row = {"weight_dtype": wname, "table_types_on": mode}
        for name, (K, N) in SHAPES.items():
            row[name] = {"M=%d" % M: round(time_linear(wname, K, N, M, group=args.group), 2) for M in (1, 4, 8)}
        print(json.dumps(row), flush=True)
    if args.engine:
        for wname in args.types.split(","):
            ms = engine_rate(wname, args.layers)
            print(json.dumps({"engine": wname, "layers": args.layers, "ms_per_token": round(ms, 4),
                              "tokens_per_s": round(1e3 / ms, 1)}), flush=True)
