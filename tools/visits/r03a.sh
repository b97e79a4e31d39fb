#!/bin/bash
# r03a: XQ GEMV probe — round-2 kernel vs streamed kernel grid vs load-only twin, plus timelines
set -u
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout 300 tools/xq_probe.bin 5 > $OUT/xq_probe.txt 2>&1
echo "probe rc=$?"
for s in qkv o gate_up down; do
  timeout 120 tools/xq_probe_stamps.bin 1 $s 8 3 2>&1 | grep -A9 timeline >> $OUT/xq_timeline.txt
  timeout 120 tools/xq_probe_stamps.bin 1 $s 4 2 2>&1 | grep -A9 timeline >> $OUT/xq_timeline.txt
done
cat $OUT/xq_probe.txt
cat $OUT/xq_timeline.txt
