#!/bin/bash
# r06g: knock-outs of the 256-row GEMM kernel (timing only): where its time is
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
for lib in base k1 k2 k4 k8 k15; do
  L=""; [ $lib != base ] && L=$PWD/tools/lib_gemm_$lib.so
  echo "== $lib"
  WOQ_HIP_LIB=$L timeout 120 python tools/gemm_one.py 8192 4096 22016 f32 2>&1 | tail -1
  WOQ_HIP_LIB=$L timeout 120 python tools/gemm_one.py 8192 11008 4096 f16 2>&1 | tail -1
done 2>&1 | tee $O/knockouts.txt
echo "== 128-row kernels"; WOQ_GEMM_TALL=0 timeout 120 python tools/gemm_one.py 8192 4096 22016 f32 2>&1 | tail -1 | tee -a $O/knockouts.txt
WOQ_GEMM_TALL=0 timeout 120 python tools/gemm_one.py 8192 11008 4096 f16 2>&1 | tail -1 | tee -a $O/knockouts.txt
