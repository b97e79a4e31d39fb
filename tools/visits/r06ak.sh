#!/bin/bash
# r06ak: the tree after the request-order change — full GPU suite, smoke, the bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ak; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"; cat $O/bench20.json; cp bench_extra.json $O/bench20_extra.json
