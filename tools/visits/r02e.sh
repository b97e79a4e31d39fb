#!/bin/bash
set -u
OUT=gpurun_out/r02e; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "logits_vs_oracle or fp32_activation or graph_replay" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $OUT/pytest1.log
B="python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
for i in 1 2; do
WOQ_GEMV_LC=0 timeout 120 $B > $OUT/bench_xq_$i.json 2>$OUT/err.txt; echo xq $(python -c "import json;d=json.load(open('$OUT/bench_xq_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
timeout 120 $B > $OUT/bench_lc_$i.json 2>$OUT/err.txt; echo lc $(python -c "import json;d=json.load(open('$OUT/bench_lc_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
done
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- $B --steps 32 --warmup 8 > $OUT/bench_prof.json 2> $OUT/rocprof.err
python tools/prof_stats.py $(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1) 12 > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
