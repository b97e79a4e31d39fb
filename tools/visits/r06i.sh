#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
for rep in 1 2; do
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  echo "== tall=$1 tall_raw=$2 rep $rep"
  WOQ_GEMM_TALL=$1 WOQ_GEMM_TALL_RAW=$2 timeout 300 python tools/prefill_engine_bench.py 4 2048 2>&1 | tail -1 | cut -c150-400
done; done 2>&1 | tee $O/engine_ab.txt
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  echo "== c2 (32 x 2048, g32 asym) tall=$1 tall_raw=$2"
  WOQ_GEMM_TALL=$1 WOQ_GEMM_TALL_RAW=$2 timeout 300 python tools/prefill_engine_bench.py 32 2048 32 1 2>&1 | tail -1 | cut -c150-400
done 2>&1 | tee -a $O/engine_ab.txt
