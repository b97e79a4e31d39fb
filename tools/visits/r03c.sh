#!/bin/bash
# r03c: lean streamed XQ GEMV (weights first, balanced digits, LDS factor tables) — grid + knock-outs
set -u
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 300 tools/xq_probe.bin 5 > $OUT/xq_probe.txt 2>&1
( timeout 300 tools/xq_probe_knobs.bin 5 qkv 8 4; timeout 300 tools/xq_probe_knobs.bin 5 o 4 3
  timeout 300 tools/xq_probe_knobs.bin 5 gate_up 8 4; timeout 300 tools/xq_probe_knobs.bin 5 down 8 3 ) > $OUT/xq_knobs.txt 2>&1
cat $OUT/xq_probe.txt $OUT/xq_knobs.txt
