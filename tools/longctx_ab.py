"""Same-box A/B of the long-context decode attention forms (round 4) on the BASELINE configs[4] shape (Mistral-7B:
32 q / 8 kv heads, fp8 KV, 8192 cached positions), `layers` layers: per variant the decode time per token after 0.8 s
of clock conditioning, and the per-layer difference against the first variant.
  python tools/longctx_ab.py [layers=16] [ctx=8192] [kv=fp8|fp16]
variants: fixed-chunk slices + in-launch merge (the default) | adaptive slices + in-launch merge | adaptive slices +
combine launch (round 3's form; WOQ_ATTN_FOLD=0 is read at engine creation) | fixed + combine launch."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intel_extension_for_transformers_amd import _lib as L  # noqa: E402
from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def run(layers, ctx, kv, fold, chunk, splits, fs=1, grouped=-1):
    """chunk: 0 = adaptive slices, else the fixed slice length; splits: 0 = tune_attn_for's choice; fs: context slices
    inside the fused qkv launch (round 6; WOQ_FUSE_SLICED is read at engine creation); grouped: -1 = tune_attn_for's
    choice, 0 / 1 = per-query-head slices / the grouped matrix-core form"""
    os.environ["WOQ_ATTN_FOLD"] = "1" if fold else "0"
    os.environ["WOQ_FUSE_SLICED"] = "1" if fs & 1 else "0"
    os.environ["WOQ_GROUPED_A2A"] = "1" if fs & 2 else "0"  # fs bit 1: grouped slices merge among themselves
    hidden, heads, hd, vocab = 4096, 32, 128, 32000
    kvh, inter = int(os.environ.get("LCAB_KVH", "8")), int(os.environ.get("LCAB_INTER", "14336"))  # 32 / 11008: Llama-2-7B
    eng = WoqDecoderEngine(hidden, inter, heads, kvh, hd, layers, vocab, max_ctx=ctx + 256,
                           kv_dtype=torch.float8_e4m3fn if kv == "fp8" else torch.float16)
    synth_llama_weights(eng, hidden, inter, heads, kvh, hd, layers, vocab, group=128, sym=True, scale_dtype="fp16")
    g = torch.Generator().manual_seed(1)
    toks = torch.randint(0, vocab, (ctx,), generator=g).cuda()
    for s0 in range(0, ctx, 2048):
        eng.prefill(toks[s0:s0 + 2048], start_pos=s0, greedy=True)
    eng.tune_attn_for(ctx + 128)
    if kvh == heads:  # multi-head: the per-query-head slices (no grouped form)
        if splits:
            eng.set_attn_splits(splits)
        chunk = splits = 0
    eng.set_attn_chunk(chunk)
    if grouped >= 0:
        eng.set_attn_grouped(bool(grouped))
    if kvh == heads:
        pass
    elif splits:
        eng.set_attn_splits(splits)
    elif chunk:
        eng.set_attn_splits(max(2, min(64, -(-(ctx + 128) // chunk))))
    else:
        eng.set_attn_splits(max(2, min(64, (ctx + 128) // 256)))
    run = eng.run  # eager bursts (the default launch mode since r04h); LCAB_GRAPH=1: the replayed graph
    if os.environ.get("LCAB_GRAPH") == "1":
        eng.capture(greedy=True)
        run = eng.replay_graph
    tok0, pos0 = eng.token.clone(), eng.pos.clone()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:  # clock conditioning
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        run(64)
        torch.cuda.synchronize()
    eng.token.copy_(tok0)
    eng.pos.copy_(pos0)
    run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(64)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 64
    reps = []
    for _ in range(3):  # how stable is the number inside one engine instance?
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        run(64)
        torch.cuda.synchronize()
        reps.append(round((time.perf_counter() - t0) / 64 * 1e3, 4))
    kptr = L.lib().woq_engine_kv_cache_ptr(eng._h, 0)
    out = dict(fs=fs, fused=eng.uses_fused_attn(), grouped=bool(L.lib().woq_engine_attn_grouped(eng._h)), fold=fold, splits=L.lib().woq_engine_attn_splits(eng._h), chunk=L.lib().woq_engine_attn_chunk(eng._h),
               ms_per_token=round(dt * 1e3, 4), again=reps, token=int(eng.token.item()), status=eng.status(),
               kcache="%x" % kptr, qkv0="%x" % eng.layer_tensors[0]["qkv"].data_ptr(),
               gu0="%x" % eng.layer_tensors[0]["gate_up"].data_ptr())
    del eng
    torch.cuda.empty_cache()
    return out


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    kv = sys.argv[3] if len(sys.argv) > 3 else "fp8"
    # variants "fold:chunk:splits" (argv[4:]); default: the forms of DESIGN.md 3.2
    variants = sys.argv[4:] or ["0:0:0", "1:0:0", "0:256:0", "1:256:0", "0:0:16", "1:0:16", "0:0:24", "0:512:0", "0:0:0"]
    base = None
    for v in variants:
        f = [int(x) for x in v.split(":")]  # fold:chunk:splits[:fs[:grouped]]
        fold, chunk, splits = f[:3]
        r = run(layers, ctx, kv, bool(fold), chunk, splits, f[3] if len(f) > 3 else 1, f[4] if len(f) > 4 else -1)
        if base is None:
            base = r["ms_per_token"]
        r["us_per_layer_vs_first"] = round((r["ms_per_token"] - base) * 1e3 / layers, 2)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
