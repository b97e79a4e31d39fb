#!/bin/bash
# r06f: kernel trace of the 4 x 2048 prompt pass, 256-row tiles on / off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
for v in 1 0; do
  WOQ_GEMM_TALL=$v timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof$v -o pf -- python tools/prefill_engine_bench.py 4 2048 > $O/pf_tall$v.txt 2> $O/rocprof$v.err; echo "rocprof tall=$v rc=$?"; tail -1 $O/pf_tall$v.txt | cut -c1-200
  python tools/prof_stats.py $(ls $O/prof$v/*.db $O/prof$v/*/*.db 2>/dev/null | head -1) 12 > $O/kernel_stats_tall$v.txt 2>&1; cat $O/kernel_stats_tall$v.txt | cut -c1-180; rm -rf $O/prof$v
done
