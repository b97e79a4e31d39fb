#!/bin/bash
# r06aq: the round's last library — full GPU suite + smoke once more
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aq; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.txt | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
