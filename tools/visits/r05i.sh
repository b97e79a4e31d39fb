#!/bin/bash
# r05i: fp8 weights at decode rows after round 5 (K = 11008 on the fp8 MFMA kernel; groups of 32), against the lookup kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
for g in 128 32; do
  timeout 200 python tools/table_decode_bench.py --types fp8_e4m3,fp8_e5m2 --group $g >> $O/fp8.txt 2>> $O/fp8.err
  WOQ_FP8_GENERIC=1 timeout 200 python tools/table_decode_bench.py --types fp8_e4m3 --group $g >> $O/fp8_generic.txt 2>> $O/fp8.err
done
echo "--- fp8 MFMA kernel (group 128 rows, then group 32)"; cat $O/fp8.txt; echo "--- lookup kernel"; cat $O/fp8_generic.txt
