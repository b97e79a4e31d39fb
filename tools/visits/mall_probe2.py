"""Development: a chain of decode GEMVs over DISTINCT weight blobs (cold every time) captured in one graph, with and
without a second stream that reads blob i + 1 while GEMV i runs."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from intel_extension_for_transformers_amd import qbits  # noqa: E402

e = torch.empty(0)
def chain(K, N, nblobs, prefetch, pf_frac=1.0):
    w = torch.randn(K, N, device="cuda") * 0.02
    b0 = qbits.quantize_to_packed_weight(w, False, 128, "fp32", "int4_clip", "fp16", False)
    blobs = [b0.clone() for _ in range(nblobs)]
    hdr = qbits.header_of(b0)
    x = torch.randn(1, K, device="cuda")
    out = torch.empty(1, N, device="cuda")
    sink = torch.zeros(nblobs, device="cuda", dtype=torch.int64)
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    def body():
        for i, b in enumerate(blobs):
            if prefetch and i + 1 < nblobs:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    nb = blobs[i + 1]
                    n4 = int(nb.numel() // 4 * pf_frac)
                    sink[i] = nb.view(torch.int32)[:n4].sum()
            qbits.woq_linear(x, b, e, out, "fp32", "int4_clip", "fp16", False)
        main.wait_stream(side)
    with torch.cuda.stream(main):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / nblobs)
    ts.sort()
    return ts[len(ts) // 2], b0.numel() / 1e6

for K, N, nb in ((4096, 22016, 12), (11008, 4096, 24), (4096, 4096, 48)):
    base, mb = chain(K, N, nb, False)
    pf, _ = chain(K, N, nb, True)
    print("K %5d N %5d (%.1f MB x %d): %.2f us / GEMV (%.2f TB/s) alone | %.2f us with blob i+1 read on a second stream"
          % (K, N, mb, nb, base, mb / base, pf))
