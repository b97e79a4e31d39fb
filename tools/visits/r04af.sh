#!/bin/bash
# perm-mask table lookup as the default: the table-type tests (tile and XQ kernels) and the engine timings again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04af; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py tests/test_gpu_api.py -q -m gpu -k "table" > $O/pytest_table.txt 2>&1
tail -2 $O/pytest_table.txt
timeout 100 python tools/table_decode_bench.py --engine --types nf4,nf4:bf16,fp4_e2m1 > $O/table_bench.txt 2> $O/table_bench.err
cat $O/table_bench.txt
