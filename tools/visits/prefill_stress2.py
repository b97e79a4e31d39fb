"""Development stress: does a prompt pass depend on what the previous pass left behind? Steady result of prompt A
repeated, then A alternated with another prompt B: every A result must equal the steady one."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests.test_gpu_fullsize_oracle import build_7b_shape  # noqa: E402

group, asym = int(sys.argv[1]), bool(int(sys.argv[2]))
eng, oracle, cfg = build_7b_shape(int(sys.argv[3]) if len(sys.argv) > 3 else 8, group, asym)
A, B = [11, 20000, 317], [5, 9999, 31000]
res = [eng.prefill(A, greedy=False)[0].clone() for _ in range(6)]
print("A repeated: equal to run 5:", [bool(torch.equal(r, res[5])) for r in res])
steady = res[5]
bad = 0
for i in range(60):
    eng.prefill(B, greedy=False)
    got = eng.prefill(A, greedy=False)[0].clone()
    if not torch.equal(got, steady):
        bad += 1
        if bad <= 3:
            print("alternating run %d: max abs diff vs steady A %.3e" % (i, float((got - steady).abs().max())))
print("alternating: %d / 60 A results differ from the steady A result" % bad)
