"""Prompt-pass measurement (BASELINE configs[2] / [4] shapes): n_seq x T tokens through the full Llama-2-7B-shaped
engine (all 32 layers: MFMA GEMMs + attention), tokens/s and model TFLOP/s against the 2.5 PFLOP/s dense fp16 peak.
args: n_seq T [group] [asym 0/1] [layers] [kv_heads] [inter] [chunk]. A development / profile tool; bench.py is the
headline."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def main():
    n_seq, T = int(sys.argv[1]), int(sys.argv[2])
    group = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    asym = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    layers = int(sys.argv[5]) if len(sys.argv) > 5 else 32
    kv_heads = int(sys.argv[6]) if len(sys.argv) > 6 else 32
    inter = int(sys.argv[7]) if len(sys.argv) > 7 else 11008
    chunk = int(sys.argv[8]) if len(sys.argv) > 8 else T
    hidden, heads, hd, vocab = 4096, 32, 128, 32000
    eng = WoqDecoderEngine(hidden, inter, heads, kv_heads, hd, layers, vocab, max_ctx=T, max_batch=n_seq)
    synth_llama_weights(eng, hidden, inter, heads, kv_heads, hd, layers, vocab, group=group, sym=not asym,
                        scale_dtype="fp16")
    g = torch.Generator().manual_seed(1234)
    toks = torch.randint(0, vocab, (n_seq, T), generator=g).cuda()

    def run():
        for s0 in range(0, T, chunk):
            eng.prefill(toks[:, s0:s0 + chunk], start_pos=s0)

    run()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    params = layers * (hidden * (heads + 2 * kv_heads) * hd + heads * hd * hidden + 3 * hidden * inter)
    lin = 2.0 * params * n_seq * T
    att = layers * n_seq * 4.0 * heads * hd * T * (T + 1) / 2  # QK^T + PV, causal
    out = dict(n_seq=n_seq, T=T, chunk=chunk, group=group, asym=asym, layers=layers, kv_heads=kv_heads, inter=inter,
               s=dt, tokens_per_s=n_seq * T / dt, linear_tflops=lin / dt / 1e12, total_tflops=(lin + att) / dt / 1e12,
               mfma_frac_of_2500=(lin + att) / dt / 1e12 / 2500.0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
