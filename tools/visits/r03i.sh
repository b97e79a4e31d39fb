#!/bin/bash
# r03i: issue order of the lean XQ GEMV — how many weight tiles go in front of the small requests (same box, 4 builds)
set -u
OUT=gpurun_out/r03i
mkdir -p $OUT
for v in "" _pre0 _pre1 _pre2; do
  echo "=== build xq_probe$v (PRE=${v:-all})" >> $OUT/xq_order.txt
  timeout 300 tools/xq_probe$v.bin 5 2>&1 | grep -E "^[a-z_]+ |r02 kernel|tpw 8|tpw 4" >> $OUT/xq_order.txt
done
cat $OUT/xq_order.txt
