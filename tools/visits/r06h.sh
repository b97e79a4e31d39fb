#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
for t in 1 0 1 0; do
  echo "== tall=$t"
  for sh in "8192 4096 4096 f16" "8192 11008 4096 f16" "8192 4096 12288 f32" "8192 4096 22016 f32" "2048 4096 22016 f32" "2048 4096 4096 f16"; do
    WOQ_GEMM_TALL=$t timeout 120 python tools/gemm_one.py $sh 2>&1 | tail -1
  done
done 2>&1 | tee $O/shapes.txt
