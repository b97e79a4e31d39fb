#!/bin/bash
set -u
OUT=gpurun_out/r02i; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "logits_vs_oracle or graph_replay or fp32_activation" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $OUT/pytest1.log
B="python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { tag=$1; shift; env "$@" timeout 120 $B > $OUT/bench_$tag.json 2>$OUT/err_$tag.txt; echo $tag $(python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print(d['value'], d['roofline']['us_per_launch'])"); }
run xq1 X=1
run f32a WOQ_ENGINE_XQ=0
run xq_tpw4 WOQ_TILE_TPW4=32
run xq2 X=1
run f32b WOQ_ENGINE_XQ=0
