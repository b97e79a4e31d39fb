#!/bin/bash
# r05a: Infinity-Cache probe of the decode launches + the token-long prefetcher's sweep, default and no-nt builds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout 300 python tools/visits/r05a_mall.py probe sweep > $O/default.txt 2> $O/default.err; echo "default rc=$?"
cat $O/default.txt; tail -3 $O/default.err
WOQ_HIP_LIB=$PWD/tools/lib_xq_nont.so timeout 300 python tools/visits/r05a_mall.py probe sweep > $O/nont.txt 2> $O/nont.err; echo "nont rc=$?"
cat $O/nont.txt; tail -3 $O/nont.err
timeout 400 python -m pytest tests/test_gpu_tp_device.py -q -m gpu -k "bench_py" > $O/pytest_bench_py.txt 2>&1
tail -3 $O/pytest_bench_py.txt
