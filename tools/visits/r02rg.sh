#!/bin/bash
# r02rg: half-tile ring form of the hand-scheduled GEMM (WOQ_GEMM_RING=1) against the two-tile form (=0)
mkdir -p gpurun_out
export PROBE_CLEAR=1 PROBE_REPEAT=6
for ring in 1 0; do
for cfg in "8192 4096 22016 128 0 f32" "8192 4096 22016 32 1 f32" "8192 11008 4096 128 0 f16" "8192 4096 12288 128 1 f32" "200 1024 1152 64 1 f16" "130 4096 384 128 0 f32" "9 512 4096 32 1 f16"; do
  set -- $cfg
  if [ $6 = f16 ]; then export PROBE_ACT16=1; else unset PROBE_ACT16; fi
  WOQ_GEMM_RING=$ring timeout 200 tools/gemm_probe.bin tools/lib_gemm_ring.so $1 $2 $3 $4 $5 bf16 10 | grep -E "TFLOP|check|never|repeat" | sed "s/tools\/lib_gemm_ring.so/ring=$ring $6/; s/ incl. pack pass//" | tr '\n' ' ' | cut -c1-330; echo
done
done 2>&1 | tee gpurun_out/r02rg.txt
