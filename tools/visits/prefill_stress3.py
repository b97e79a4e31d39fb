"""Development: is the deviant prompt pass tied to an idle (down-clocked) GPU?"""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests.test_gpu_fullsize_oracle import build_7b_shape  # noqa: E402

eng, oracle, cfg = build_7b_shape(8, 128, False)
A = [11, 20000, 317]
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
def busy(sec):
    t0 = time.time()
    while time.time() - t0 < sec:
        for _ in range(20):
            (x @ x)
        torch.cuda.synchronize()
busy(2.0)
res = [eng.prefill(A, greedy=False)[0].clone() for _ in range(8)]
steady = res[-1]
print("after 2 s of GPU work: equal to steady:", [bool(torch.equal(r, steady)) for r in res])
for idle in (0.5, 2.0, 5.0):
    torch.cuda.synchronize()
    time.sleep(idle)
    res = [eng.prefill(A, greedy=False)[0].clone() for _ in range(6)]
    print("after %.1f s idle: equal to steady:" % idle, [bool(torch.equal(r, steady)) for r in res])
busy(1.0)
res = [eng.prefill(A, greedy=False)[0].clone() for _ in range(6)]
print("after 1 s busy again: equal to steady:", [bool(torch.equal(r, steady)) for r in res])
