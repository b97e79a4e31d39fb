import sys, time, torch
sys.path.insert(0, ".")
from intel_extension_for_transformers_amd import qbits
K, N, M = 4096, 22016, 8192
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.randn(K, N, device="cuda", generator=g) * 0.02
x = torch.randn(M, K, device="cuda", generator=g)
e = torch.empty(0)
for wt in ("nf4", "fp8_e4m3"):
    for comp in ("fp32", "bf16"):
        blob = qbits.quantize_to_packed_weight(w, False, 128, comp, wt, "fp32", False)
        deq = torch.empty(K, N, device="cuda")
        qbits.dequantize_packed_weight(blob, deq, False, comp, wt, "fp32")
        out = torch.empty(M, N, device="cuda")
        qbits.woq_linear(x, blob, e, out, comp, wt, "fp32", False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            qbits.woq_linear(x, blob, e, out, comp, wt, "fp32", False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        rows = [0, 127, 4096, 8191]
        ref = x[rows].double() @ deq.double()
        err = ((out[rows].double() - ref).abs().max(dim=1).values / ref.abs().max(dim=1).values).max().item()
        print(f"{wt} compute {comp}: {ms:.2f} ms  {2.0*M*K*N/ms/1e9:.0f} TFLOP/s  worst row err / rowmax {err:.2e}", flush=True)
        del blob, deq, out
