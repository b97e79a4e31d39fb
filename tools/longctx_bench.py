"""BASELINE configs[4] shape: Mistral-7B (hidden 4096, inter 14336, 32 q / 8 kv heads, 32 layers) int4 sym g128 with
an fp8 (e4m3) KV cache: chunked prefill of an 8k-token prompt, then batch-1 decode at that context.
args: [ctx=8192] [chunk=2048] [kv=fp8|fp16] [splits=0] [group=128] [asym=0] [inter=14336] [kv_heads=8] [grouped=0] — splits 0 =
the engine's own choice (tune_attn_for), else forced, with the grouped-query slices iff grouped=1; args 5-8 turn
it into any Llama-class shape (e.g. configs[2]: `160 160 fp16 0 32 1 11008 32`). Development / profile tool (bench.py
is the headline)."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import _lib as L  # noqa: E402
from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def main():
    ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    kv = sys.argv[3] if len(sys.argv) > 3 else "fp8"
    splits = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    group = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    asym = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
    inter = int(sys.argv[7]) if len(sys.argv) > 7 else 14336
    kvh = int(sys.argv[8]) if len(sys.argv) > 8 else 8
    grouped = bool(int(sys.argv[9])) if len(sys.argv) > 9 else False
    hidden, heads, hd, layers, vocab = 4096, 32, 128, 32, 32000
    eng = WoqDecoderEngine(hidden, inter, heads, kvh, hd, layers, vocab, max_ctx=ctx + 256,
                           kv_dtype=torch.float8_e4m3fn if kv == "fp8" else torch.float16, attn_splits=max(splits, 0) or 1)
    synth_llama_weights(eng, hidden, inter, heads, kvh, hd, layers, vocab, group=group, sym=not asym, scale_dtype="fp16")
    g = torch.Generator().manual_seed(1)
    toks = torch.randint(0, vocab, (ctx,), generator=g).cuda()

    def prefill():
        for s0 in range(0, ctx, chunk):
            eng.prefill(toks[s0:s0 + chunk], start_pos=s0, greedy=True)

    prefill()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prefill()
    torch.cuda.synchronize()
    t_pf = time.perf_counter() - t0
    if splits == 0:
        eng.tune_attn_for(ctx + 128)
    else:
        eng.set_attn_grouped(grouped)
    eng.capture(greedy=True)
    eng.replay(8)
    torch.cuda.synchronize()
    n = 64
    t0 = time.perf_counter()
    eng.replay(n)
    torch.cuda.synchronize()
    t_dec = (time.perf_counter() - t0) / n
    params = layers * (hidden * (heads + 2 * kvh) * hd + heads * hd * hidden + 3 * hidden * inter)
    wbytes = params // 2 + params // group * 2 + (params // group // 2 if asym else 0)
    kv_bytes = 2 * layers * kvh * hd * ctx * (1 if kv == "fp8" else 2)
    print(json.dumps(dict(ctx=ctx, chunk=chunk, kv=kv, splits=L.lib().woq_engine_attn_splits(eng._h),
                          grouped=bool(L.lib().woq_engine_attn_grouped(eng._h)), group=group, asym=asym, inter=inter, kv_heads=kvh, prefill_s=t_pf, prefill_tok_s=ctx / t_pf,
                          decode_ms=t_dec * 1e3, decode_tok_s=1 / t_dec,
                          decode_gbps_weights=wbytes / t_dec / 1e9, decode_gbps_weights_plus_kv=(wbytes + kv_bytes) / t_dec / 1e9,
                          kv_mb_per_token=kv_bytes / 1e6)))


if __name__ == "__main__":
    main()
