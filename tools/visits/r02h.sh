#!/bin/bash
set -u
OUT=gpurun_out/r02h; mkdir -p $OUT
B="python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { tag=$1; shift; env "$@" timeout 120 $B > $OUT/bench_$tag.json 2>$OUT/err_$tag.txt; echo $tag $(python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print(d['value'], d['roofline']['us_per_launch'])"); }
run full X=1
run nomath WOQ_LC_DEBUG=1
run nowait WOQ_LC_DEBUG=2
run nomath_nowait WOQ_LC_DEBUG=3
run noload WOQ_LC_DEBUG=4
run noload_nomath WOQ_LC_DEBUG=5
