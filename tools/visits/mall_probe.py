"""Development: does a read by ANOTHER kernel leave a weight blob in the Infinity Cache, and how fast is the decode GEMV
from there? gate/up (45 MB) and down (22.5 MB) shapes of Llama-2-7B, M = 1."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402

def t_us(fn, pre, reps=30):
    ts = []
    for _ in range(reps):
        pre()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]

junk = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
e = torch.empty(0)
for K, N in ((4096, 22016), (11008, 4096), (4096, 12288), (4096, 4096)):
    w = torch.randn(K, N, device="cuda") * 0.02
    blob = qbits.quantize_to_packed_weight(w, False, 128, "fp32", "int4_clip", "fp16", False)
    del w
    x = torch.randn(1, K, device="cuda")
    out = torch.empty(1, N, device="cuda")
    fn = lambda: qbits.woq_linear(x, blob, e, out, "fp32", "int4_clip", "fp16", False)
    for _ in range(5): fn()
    flush = lambda: junk.add_(1)
    def flush_then_touch():
        junk.add_(1)
        blob.view(torch.int32)[: blob.numel() // 4].sum()          # another kernel reads every byte of the blob
    def warm():
        fn()
    cold = t_us(fn, flush)
    pref = t_us(fn, flush_then_touch)
    hot = t_us(fn, warm)
    mb = blob.numel() / 1e6
    print("K %5d N %5d (%.1f MB): cold %.2f us (%.2f TB/s) | read by another kernel first %.2f us (%.2f TB/s) | same kernel just ran %.2f us (%.2f TB/s)"
          % (K, N, mb, cold[0], mb / cold[0], pref[0], mb / pref[0], hot[0], mb / hot[0]))
