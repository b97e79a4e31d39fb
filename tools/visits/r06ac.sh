#!/bin/bash
# r06ac: fp8 engine with SiLU * mul formed inside the down projection's row staging — parity, timing; FETCH_SIZE of the
# fused launch at long context on 4-layer engines (the 32-layer counter passes of r06ab ran out of time)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ac; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_wtypes_decode.py tests/test_gpu_parity.py -m gpu -x -q -k "fp8" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
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
tail -3 $O/fp8_engine.txt
LCAB_GRAPH=1 timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc8 -- python tools/longctx_ab.py 4 8192 fp8 0:0:16:1:0 > $O/pmc8.json 2> $O/pmc8.err; echo "pmc8 rc=$?"
python tools/pmc_summary.py $O/pmc8 gemv_xqs > $O/lc_mistral_8k_pmc.txt 2>&1; cut -c1-200 $O/lc_mistral_8k_pmc.txt; rm -rf $O/pmc8
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc7 -- python tools/longctx_ab.py 4 2048 fp16 0:0:8:1 > $O/pmc7.json 2> $O/pmc7.err; echo "pmc7 rc=$?"
python tools/pmc_summary.py $O/pmc7 gemv_xqs > $O/lc_7b_2048_pmc.txt 2>&1; cut -c1-200 $O/lc_7b_2048_pmc.txt; rm -rf $O/pmc7
