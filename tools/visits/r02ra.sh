#!/bin/bash
# r02ra: raw-A form of the hand-scheduled GEMM (fp16 activation rows, no pack pass) against the packed form
mkdir -p gpurun_out
export PROBE_ACT16=1 PROBE_CLEAR=1 PROBE_REPEAT=8
for raw in 1 0; do
for cfg in "8192 4096 4096 128 0" "8192 11008 4096 128 0" "8192 11008 4096 32 1" "8192 4096 22016 32 1" "200 1024 1152 64 1" "130 4096 384 128 0" "9 4096 4096 128 1"; do
  WOQ_GEMM_RAW_A=$raw timeout 200 tools/gemm_probe.bin tools/lib_gemm_raw.so $cfg bf16 10 | grep -E "TFLOP|check|never|repeat" | sed "s/tools\/lib_gemm_raw.so/raw=$raw/" | tr '\n' ' '; echo
done
done 2>&1 | tee gpurun_out/r02ra.txt
