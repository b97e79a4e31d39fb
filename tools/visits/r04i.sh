#!/bin/bash
# r04i: TP rank shapes with eager bursts; bench.py multi-rank plumbing test; GPU suite (all) on the new default
set -u
TAG=r04i; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/tp_shard_bench.py 80 > $OUT/tp_shard.txt 2>&1; tail -2 $OUT/tp_shard.txt
timeout 900 python -m pytest tests/test_gpu_tp_device.py -q -k bench_py > $OUT/pytest_bench_tp.log 2>&1; echo "pytest bench tp rc=$?"; tail -8 $OUT/pytest_bench_tp.log
timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; tail -20 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
