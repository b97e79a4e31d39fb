#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel-trace of the same bench command.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [skip-tests]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/rocprof.err
echo "rocprof rc=$?"; cat $OUT/bench_prof.json
python tools/prof_stats.py $(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1) 14 > $OUT/kernel_stats.txt 2>&1
cat $OUT/kernel_stats.txt
ls $OUT/prof | head
