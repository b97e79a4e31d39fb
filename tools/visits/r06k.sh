#!/bin/bash
# r06k: rocprofv3 --pmc FETCH_SIZE over bench.py's own headline path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
export TMPDIR=/tmp
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python bench.py --gpus 1 --steps 20 --warmup 5 --counters-run > $O/pmc_bench.json 2> $O/pmc.err; echo "pmc rc=$?"; cat $O/pmc_bench.json; tail -3 $O/pmc.err | cut -c1-200
python tools/pmc_traffic.py $O/pmc $O/pmc_traffic.json gemv_xqs > $O/pmc_summary.txt 2>&1; tail -14 $O/pmc_summary.txt | cut -c1-220; rm -rf $O/pmc
