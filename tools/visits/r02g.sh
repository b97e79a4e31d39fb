#!/bin/bash
set -u
OUT=gpurun_out/r02g; mkdir -p $OUT
B="python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { tag=$1; shift; env "$@" timeout 120 $B > $OUT/bench_$tag.json 2>$OUT/err_$tag.txt; echo $tag $(python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print(d['value'], d['roofline']['us_per_launch'])"); }
run d32 X=1
run d60 WOQ_HIP_LIB=$PWD/tools/lib_d60.so
run d60nont WOQ_HIP_LIB=$PWD/tools/lib_d60nont.so
run d16 WOQ_HIP_LIB=$PWD/tools/lib_d16.so
run xq WOQ_GEMV_LC=0
