#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel-trace of the same bench command.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [tests|notests] [bench|nobench] [prof|noprof]
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-tests}" = "tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -30 $OUT/pytest_gpu.log
fi
if [ "${3:-bench}" = "bench" ]; then
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
  echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
fi
if [ "${4:-prof}" = "prof" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity > $OUT/bench_prof.json 2> $OUT/rocprof.err
  echo "rocprof rc=$?"; cat $OUT/bench_prof.json
  python tools/prof_stats.py $(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1) 14 > $OUT/kernel_stats.txt 2>&1
  cat $OUT/kernel_stats.txt
fi
