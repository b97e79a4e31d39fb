#!/bin/bash
# r02y: validation of the hand-scheduled prefill GEMM: cold process starts, engine prompt-pass repeats, the GPU tests,
# timing against hipcc's schedule
mkdir -p gpurun_out
( PROCS=30 bash tools/visits/r02w.sh cur | grep "process starts"
  timeout 600 python tools/visits/prefill_stress.py 128 0 200 8 junk 2>&1 | grep -E "differ"
  timeout 600 python tools/visits/prefill_stress.py 32 1 200 8 junk 2>&1 | grep -E "differ"
  FULL=1 bash tools/visits/r02s.sh old cur | grep TFLOP | sed 's/tools\/lib_gemm_//'
  timeout 1400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -5 ) 2>&1 | tee gpurun_out/r02y.txt
