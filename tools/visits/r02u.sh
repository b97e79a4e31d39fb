#!/bin/bash
# r02u: prefill GEMM probe, all blob variants x shapes for one or more library variants (checks included)
mkdir -p gpurun_out
for lib in "$@"; do
  for cfg in "8192 4096 22016 128 0" "8192 4096 22016 32 1" "8192 4096 22016 128 1" "8192 4096 22016 32 0" "8192 11008 4096 128 0" "8192 11008 4096 32 1" "8192 4096 12288 128 0" "4096 4096 4096 64 1" "300 4096 4096 128 0" "8192 4096 22016 4096 0"; do
    timeout 120 tools/gemm_probe.bin tools/lib_gemm_$lib.so $cfg bf16 10
  done
done 2>&1 | sed 's/tools\/lib_gemm_//' | tee gpurun_out/r02u.txt
