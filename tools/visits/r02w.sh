#!/bin/bash
mkdir -p gpurun_out
export WOQ_GEMV_AS_GEMM=1 PROBE_CLEAR=1 PROBE_REPEAT=6 PROBE_FIRST=1
for lib in "$@"; do
  n=0; bad=0
  for i in $(seq 1 ${PROCS:-40}); do
    for cfg in "3 4096 22016 128 0" "200 4096 12288 128 0"; do
      o=$(timeout 60 tools/gemm_probe.bin tools/lib_gemm_$lib.so $cfg bf16 1 2>&1)
      n=$((n+1)); if echo "$o" | grep -q "never stored\|first launch differs"; then bad=$((bad+1)); fi
    done
  done
  echo "$lib: $bad of $n process starts had unwritten / deviating outputs"
done 2>&1 | tee gpurun_out/r02w.txt
