"""Development stress: the 3-token prompt pass of tests/test_gpu_fullsize_oracle.py repeated; every result must be
bit-identical to the first (the kernels are deterministic), and other stream work is interleaved to move the timing."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests.test_gpu_fullsize_oracle import build_7b_shape  # noqa: E402

group, asym, reps = int(sys.argv[1]), bool(int(sys.argv[2])), int(sys.argv[3])
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 8
mode = sys.argv[5] if len(sys.argv) > 5 else "junk"
eng, oracle, cfg = build_7b_shape(layers, group, asym)
toks = [11, 20000, 317]
ref = None
bad = 0
junk = torch.empty(64 << 20, device="cuda")
for i in range(reps):
    if mode == "junk" and i % 3 == 1:
        junk.normal_()  # dirty caches / shift timing
    if mode != "nosync" and i % 5 == 2:
        torch.cuda.synchronize()
    got = eng.prefill(toks, greedy=False)[0].clone()
    if ref is None:
        ref = got
        o = None
        for j, t in enumerate(toks):
            o = oracle.forward_token(t, j)
        print("first vs oracle: %.3e (max %.3e)" % (float(np.abs(ref.cpu().numpy() - o).max()), float(np.abs(o).max())))
        print("first: sum %.9e l[0..3] %s" % (float(ref.double().sum()), ref[:4].tolist()))
    elif i == reps - 1 or i == 7:
        print("run %d: sum %.9e l[0..3] %s equal_first=%s" % (i, float(got.double().sum()), got[:4].tolist(), torch.equal(got, ref)))
    if ref is not got and not torch.equal(got, ref):
        bad += 1
        if bad <= 5:
            print("rep %d differs: max abs diff %.3e" % (i, float((got - ref).abs().max())))
print("g%d asym=%d layers=%d mode=%s: %d / %d repeats differ" % (group, asym, layers, mode, bad, reps))
