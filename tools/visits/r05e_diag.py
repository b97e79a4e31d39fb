import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_engine as T
for sub in (("qkv",), ("o",), ("gate_up",), ("down",)):
    eng, oracle, cfg = T._tiny(128, False, "fp16", seed=11, act_order=sub)
    prompt = [3, 17, 200]
    oracle.reset()
    got = eng.prefill(prompt)[0].cpu().numpy(); ref = oracle.forward_prompt(prompt)
    print("act-order on", sub, "prefill err", float(np.abs(got - ref).max()), "of", float(np.abs(ref).max()), flush=True)
