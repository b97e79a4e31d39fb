#!/bin/bash
# fp8 weights at decode row counts on the fp8 matrix cores (csrc/woq_gemv_fp8.hip): parity, then timings against the
# lookup kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ah; mkdir -p $O
timeout 80 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp8" > $O/pytest_fp8.txt 2>&1
tail -12 $O/pytest_fp8.txt | cut -c1-300
timeout 25 python tools/table_decode_bench.py --types fp8_e4m3,fp8_e5m2 > $O/bench_fp8_mfma.txt 2> $O/bench_fp8_mfma.err
cat $O/bench_fp8_mfma.txt; tail -2 $O/bench_fp8_mfma.err
WOQ_FP8_GENERIC=1 timeout 25 python tools/table_decode_bench.py --types fp8_e4m3 > $O/bench_fp8_lookup.txt 2> $O/bench_fp8_lookup.err
cat $O/bench_fp8_lookup.txt
