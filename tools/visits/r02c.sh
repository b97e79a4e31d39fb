#!/bin/bash
set -u
OUT=gpurun_out/r02c; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize_oracle.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
B="python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
for i in 1 2; do
WOQ_ENGINE_XQ=0 timeout 300 $B > $OUT/bench_f32_$i.json 2>$OUT/err.txt; echo f32 $(python -c "import json;d=json.load(open('$OUT/bench_f32_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
timeout 300 $B > $OUT/bench_xq_$i.json 2>$OUT/err.txt; echo xq $(python -c "import json;d=json.load(open('$OUT/bench_xq_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
done
timeout 300 python bench.py --no-extra --no-cpu-baseline --prefill-seqs 0 > $OUT/bench_parity.json 2>$OUT/err2.txt; python -c "import json;d=json.load(open('$OUT/bench_parity.json'));print(d['parity'])"
