#!/bin/bash
# r06w: full GPU test suite on the tree with self-merging slices + the bench line with its long-context rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06w; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -16 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench.err
echo "bench rc=$?"; cat $O/bench_steps20.json; cp bench_extra.json $O/bench_steps20_extra.json 2>/dev/null; tail -3 $O/bench.err
