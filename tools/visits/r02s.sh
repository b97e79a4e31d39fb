#!/bin/bash
# r02s: prefill GEMM probe over library variants (tools/lib_gemm_*.so): gate/up shape M=8192, g128 sym (+ g32 asym with FULL=1)
mkdir -p gpurun_out
for lib in "$@"; do
  timeout 120 tools/gemm_probe.bin tools/lib_gemm_$lib.so 8192 4096 22016 128 0 bf16 20
  [ -n "$FULL" ] && timeout 120 tools/gemm_probe.bin tools/lib_gemm_$lib.so 8192 4096 22016 32 1 bf16 20
done 2>&1 | tee gpurun_out/r02s.txt
