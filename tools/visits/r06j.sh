#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
WOQ_HIP_LIB=$PWD/tools/lib_gemm_split.so timeout 600 python -m pytest tests/test_gpu_fullsize_oracle.py -q -m gpu -k "prefill_gemm" --maxfail=5 2>&1 | tail -3 | cut -c1-200
for rep in 1 2; do
  for lib in base split; do
    L=""; [ $lib != base ] && L=$PWD/tools/lib_gemm_$lib.so
    echo "== $lib rep $rep"
    WOQ_HIP_LIB=$L WOQ_GEMM_TALL_RAW=0 timeout 300 python tools/prefill_engine_bench.py 4 2048 2>&1 | tail -1 | cut -c150-400
    WOQ_HIP_LIB=$L timeout 120 python tools/gemm_one.py 8192 4096 22016 f32 2>&1 | tail -1
  done
done 2>&1 | tee $O/split_ab.txt
