#!/bin/bash
set -u
OUT=gpurun_out/r02d; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "logits_vs_oracle or fp32_activation or graph_replay" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $OUT/pytest1.log
B="python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
for i in 1 2; do
WOQ_GEMV_LC=0 timeout 120 $B > $OUT/bench_xq_$i.json 2>$OUT/err.txt; echo xq $(python -c "import json;d=json.load(open('$OUT/bench_xq_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
timeout 120 $B > $OUT/bench_lc_$i.json 2>$OUT/err.txt; echo lc $(python -c "import json;d=json.load(open('$OUT/bench_lc_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
done
timeout 300 python bench.py --no-extra --no-cpu-baseline --prefill-seqs 0 > $OUT/bench_parity.json 2>$OUT/err2.txt; python -c "import json;d=json.load(open('$OUT/bench_parity.json'));print(d['parity'])"; tail -3 $OUT/err2.txt
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize_oracle.py -m gpu -q -x > $OUT/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -5 $OUT/pytest2.log
