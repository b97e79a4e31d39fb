#!/bin/bash
# r06aa: fp8-weight layers inside the decode engine — parity tests, a first 7B-shape timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_wtypes_decode.py tests/test_gpu_parity.py -m gpu -x -q -k "fp8" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout 600 python - > $O/fp8_engine.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
for wd in ("fp8_e4m3", "int4_clip"):
    eng = bench.build_engine(bench.LLAMA2_7B, group=128, sym=True, max_ctx=512, weight_dtype=wd)
    bench.feed_prompt(eng, bench.LLAMA2_7B["vocab"], 32)
    eng.capture(greedy=True)
    eng.replay_graph(8); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.replay_graph(64); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 64
    print(wd, "ms/token %.4f tokens/s %.1f xq=%s status=%d" % (dt * 1e3, 1 / dt, eng.uses_xq(), eng.status()))
    del eng; bench.free_gpu()
PY
cat $O/fp8_engine.txt | tail -5
