#!/bin/bash
# r04n: sampling + repetition penalty on the engine (device sampler): API tests, sampled-decode rate
set -u
TAG=r04n; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py -q > $OUT/pytest_api.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_api.log
timeout 600 python - > $OUT/sampled_rate.txt 2>&1 <<'PY'
import sys, time, json
import torch
sys.path.insert(0, ".")
from bench import LLAMA2_7B, build_engine
from intel_extension_for_transformers_amd.runtime.engine import DeviceSampler, generate_sampled
eng = build_engine(LLAMA2_7B, max_ctx=1024)
prompt = torch.randint(0, 32000, (32,)).tolist()
for name, smp in (("reference NeuralChat defaults (do_sample, T 0.1, top_k 40, top_p 0.75, penalty 1.1)", DeviceSampler(True, 0.1, 40, 0.75, 1.1)),
                  ("repetition penalty only (argmax)", DeviceSampler(False, 1.0, 0, 1.0, 1.1))):
    generate_sampled(eng, prompt, 16, smp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = generate_sampled(eng, prompt, 256, smp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"request": name, "new_tokens": len(out), "tokens_per_s_incl_prompt_pass": len(out) / dt, "ms_per_token": dt / len(out) * 1e3}))
ids = eng.generate(prompt, 16)
torch.cuda.synchronize()
t0 = time.perf_counter()
ids = eng.generate(prompt, 256)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"request": "engine.generate greedy (per-token host read)", "tokens_per_s_incl_prompt_pass": 256 / dt}))
PY
cat $OUT/sampled_rate.txt | grep -v amdgpu.ids
