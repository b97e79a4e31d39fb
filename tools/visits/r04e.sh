#!/bin/bash
# r04e: slice-count sweep of the long-context decode attention (grouped fp8 at 8k; multi-head fp16 at 2048 / 512), kernel trace at the best
set -u
TAG=r04e; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/longctx_ab.py 16 8192 fp8 0:0:32 0:0:28 0:0:26 0:0:24 0:0:23 0:0:22 0:0:20 0:0:16 0:0:12 0:0:24 0:0:32 > $OUT/sweep_gq_8k.txt 2>&1; cat $OUT/sweep_gq_8k.txt
timeout 600 python tools/longctx_ab.py 16 4096 fp8 0:0:16 0:0:12 0:0:11 0:0:10 0:0:8 0:0:16 > $OUT/sweep_gq_4k.txt 2>&1; cat $OUT/sweep_gq_4k.txt
LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 2048 fp16 0:0:32 0:0:16 0:0:8 0:0:6 0:0:4 0:0:32 > $OUT/sweep_mha_2k.txt 2>&1; cat $OUT/sweep_mha_2k.txt
LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 512 fp16 0:0:8 0:0:4 0:0:3 0:0:2 0:0:8 > $OUT/sweep_mha_512.txt 2>&1; cat $OUT/sweep_mha_512.txt
for v in "0:0:24"; do
  n=$(echo $v | tr ':' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$n -o lc -- python tools/longctx_ab.py 8 8192 fp8 $v > $OUT/prof_$n.txt 2>&1
  DB=$(ls $OUT/prof_$n/*.db $OUT/prof_$n/*/*.db 2>/dev/null | head -1)
  echo "== variant $v"; python tools/prof_stats.py $DB 8 2>&1 | cut -c1-60,92-170 | tee $OUT/kernels_$n.txt
  rm -rf $OUT/prof_$n
done
