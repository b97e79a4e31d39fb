#!/bin/bash
# r05y: PMC traffic pass (FETCH_SIZE) of the decode GEMVs on the round's last tree; second attempt with a lighter driver
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures --condition-ms 0 > $GRAFT_REPO_ROOT/$O/pmc_bench.json 2> $GRAFT_REPO_ROOT/$O/pmc.err; echo "pmc rc=$?"
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $O/pmc $O/pmc_traffic.json gemv_xqs > $O/pmc_summary.txt 2>&1; tail -4 $O/pmc_summary.txt | cut -c1-220
if [ ! -f $O/pmc_traffic.json ]; then
  rm -rf $O/pmc; cd /tmp
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/tools/visits/r05y_drv.py > $GRAFT_REPO_ROOT/$O/pmc_dev.json 2> $GRAFT_REPO_ROOT/$O/pmc_dev.err; echo "pmc(dev_bench) rc=$?"
  cd $GRAFT_REPO_ROOT
  python tools/pmc_traffic.py $O/pmc $O/pmc_traffic.json gemv_xqs > $O/pmc_summary.txt 2>&1; tail -4 $O/pmc_summary.txt | cut -c1-220
fi
rm -rf $O/pmc
