#!/bin/bash
# r03b: knock-outs of the streamed XQ GEMV (which stage costs what) + twin with limb slices
set -u
OUT=gpurun_out/r03b
mkdir -p $OUT
for cfg in "8 4" "4 2" "8 99"; do
  timeout 300 tools/xq_probe_knobs.bin 5 all_shapes_dummy $cfg > /dev/null 2>&1
done
( timeout 300 tools/xq_probe_knobs.bin 5 qkv 8 4; timeout 300 tools/xq_probe_knobs.bin 5 qkv 4 3; timeout 300 tools/xq_probe_knobs.bin 5 qkv 8 99
  timeout 300 tools/xq_probe_knobs.bin 5 o 4 3; timeout 300 tools/xq_probe_knobs.bin 5 o 2 99
  timeout 300 tools/xq_probe_knobs.bin 5 gate_up 8 4; timeout 300 tools/xq_probe_knobs.bin 5 gate_up 4 99
  timeout 300 tools/xq_probe_knobs.bin 5 down 8 3 ) > $OUT/xq_knobs.txt 2>&1
cat $OUT/xq_knobs.txt
