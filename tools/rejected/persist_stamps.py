"""Where a projection's time goes inside the persistent launch (csrc/woq_persist.hip PS_STAMP): per projection kind,
averaged over workgroups and layers >= 2, in us (100 MHz wall clock). usage: persist_stamps.py [layers]"""
import sys

import torch

sys.path.insert(0, ".")
from intel_extension_for_transformers_amd.runtime import WoqDecoderEngine, synth_llama_weights  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    eng = WoqDecoderEngine(4096, 11008, 32, 32, 128, layers, 32000, max_ctx=512)
    synth_llama_weights(eng, 4096, 11008, 32, 32, 128, layers, 32000, group=128, sym=True, scale_dtype="fp16")
    eng.set_persist(True)
    assert eng.uses_persist()
    eng.reset(1, 0)
    for _ in range(20):
        eng.step(True)
    st, ring = eng.persist_stamps(True)
    eng.step(True)
    torch.cuda.synchronize()
    s = st.cpu().double() / 100.0  # us
    G = s.shape[0]
    print(f"grid {G}, ring {ring} tiles, status {eng.status()}")
    t0 = s[:, :, 0][s[:, :, 0] > 0].min()
    names = ["qkv", "o", "gate/up", "down"]
    print("consumer wave 0 (mean over workgroups, layers >= 2); columns: wait hints | sweep | meet1 | tiles | meet2 | "
          "epilogue | (attention) || op total || loader: issue span, stalls, stall us")
    for k in range(4):
        rows = []
        for l in range(2, layers):
            x = s[:, l * 4 + k, :]
            rows.append(torch.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 4] - x[:, 3],
                                     x[:, 5] - x[:, 4], x[:, 6] - x[:, 5], (x[:, 7] - x[:, 6]) if k == 0 else x[:, 6] * 0,
                                     x[:, 6] - x[:, 0], x[:, 9] - x[:, 8], st[:, l * 4 + k, 10].cpu().double(),
                                     st[:, l * 4 + k, 11].cpu().double() / 100.0], 1))
        r = torch.stack(rows).mean(0)  # [G, 11]
        m, mx = r.mean(0), r.max(0).values
        print(f"{names[k]:8s} " + " ".join(f"{v:6.2f}" for v in m.tolist()))
        print(f"{'  (max)':8s} " + " ".join(f"{v:6.2f}" for v in mx.tolist()))
    # the layer's timeline for workgroup 0 and the slowest starter, layer 3
    l = min(3, layers - 1)
    for wg in (0, 17, 255):
        base = s[wg, l * 4, 0]
        line = []
        for k in range(4):
            x = s[wg, l * 4 + k]
            line.append(f"{names[k]}: " + " ".join(f"{(v - base):6.2f}" for v in x[:10].tolist()) + "  | loader waits: " +
                        " ".join(f"{(v - base):6.2f}" for v in x[12:16].tolist()) + f" | tile passes: waiting {st[wg, l * 4 + k, 16].item() / 100.0:5.2f} working {st[wg, l * 4 + k, 17].item() / 100.0:5.2f}")
        print(f"wg {wg} layer {l} (us from the qkv start): \n   " + "\n   ".join(line))
    # spread of op-start across workgroups (how synchronous the chip is)
    for k in range(4):
        x = s[:, l * 4 + k, 0]
        print(f"{names[k]} start spread over workgroups: {(x.max() - x.min()):.2f} us; end spread "
              f"{(s[:, l * 4 + k, 6].max() - s[:, l * 4 + k, 6].min()):.2f} us")
    span = s[:, (layers - 1) * 4 + 3, 6].max() - s[:, 2 * 4, 0].min()
    print(f"layers 2..{layers - 1}: {span / (layers - 2):.2f} us per layer")


if __name__ == "__main__":
    main()
