#!/bin/bash
# r06e: 256-row workgroup tiles of the prefill GEMM (csrc/woq_gemm_f16t.h): parity, then same-box A/B of the prompt pass
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py -q -m gpu -k "prefill_gemm" --maxfail=8 > $O/pytest_gemm.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gemm.txt | cut -c1-250
i=0
for v in 1 0 1 0; do
  i=$((i+1))
  WOQ_GEMM_TALL=$v timeout 300 python tools/prefill_engine_bench.py 4 2048 > $O/pf_${i}_tall${v}.txt 2>&1; echo "tall=$v rc=$?"; tail -1 $O/pf_${i}_tall${v}.txt | cut -c1-300
done
WOQ_GEMM_TALL=1 timeout 300 python tools/prefill_engine_bench.py 32 2048 32 1 > $O/pf_c2_tall1.txt 2>&1; tail -1 $O/pf_c2_tall1.txt | cut -c1-300
WOQ_GEMM_TALL=0 timeout 300 python tools/prefill_engine_bench.py 32 2048 32 1 > $O/pf_c2_tall0.txt 2>&1; tail -1 $O/pf_c2_tall0.txt | cut -c1-300
