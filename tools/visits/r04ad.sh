#!/bin/bash
# nf4 / fp4 decode kernels: parity tests, then timings (digit-plane MFMA kernel vs the fp32 VALU kernel vs int4)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ad
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "table_weight" > gpurun_out/r04ad/pytest_parity.txt 2>&1
tail -5 gpurun_out/r04ad/pytest_parity.txt
timeout 400 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py -q -m gpu -k "table" > gpurun_out/r04ad/pytest_engine.txt 2>&1
tail -5 gpurun_out/r04ad/pytest_engine.txt
timeout 300 python tools/table_decode_bench.py --engine > gpurun_out/r04ad/bench_mfma.txt 2> gpurun_out/r04ad/bench_mfma.err
cat gpurun_out/r04ad/bench_mfma.txt; tail -3 gpurun_out/r04ad/bench_mfma.err
WOQ_TABLE_GENERIC=1 timeout 200 python tools/table_decode_bench.py --types nf4 > gpurun_out/r04ad/bench_generic.txt 2> gpurun_out/r04ad/bench_generic.err
cat gpurun_out/r04ad/bench_generic.txt
