#!/bin/bash
# r05n: lm_head with / without the first-row preload, kernel trace of a short decode run each (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
export TMPDIR=/tmp
for v in pre nopre pre nopre; do
  if [ $v = nopre ]; then export WOQ_HIP_LIB=$PWD/tools/lib_lm_nopre.so; else unset WOQ_HIP_LIB; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o t -- python tools/visits/r05y_drv.py > /dev/null 2>&1
  echo "== $v"; python tools/prof_stats.py $(ls $O/prof_$v/*.db $O/prof_$v/*/*.db 2>/dev/null | head -1) 12 2>&1 | grep -E "lm_head|argmax|embed" | cut -c1-40,92-170; rm -rf $O/prof_$v
done
