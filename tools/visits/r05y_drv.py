"""Lighter driver for the PMC traffic pass: an 8-layer Llama-2-7B-shaped engine, eager steps only (no hipGraph)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

cfg = dict(bench.LLAMA2_7B, layers=8)
eng = bench.build_engine(cfg, max_ctx=512, layers=8)
bench.feed_prompt(eng, cfg["vocab"], 8)
eng.run(8)
torch.cuda.synchronize()
print("done", eng.status())
