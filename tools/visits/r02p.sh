#!/bin/bash
set -u
OUT=gpurun_out/r02p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize_oracle.py -m gpu -q -x > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -4 $OUT/pytest1.log
B="python bench.py --no-extra --no-cpu-baseline --prefill-seqs 0"
run() { tag=$1; shift; env "$@" timeout 200 $B > $OUT/bench_$tag.json 2>$OUT/err_$tag.txt; echo $tag $(python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print(round(d['value'],1), round(d['roofline']['us_per_launch'],3), d['parity']['logits_max_abs'], d['parity']['greedy_tokens_equal'])"); }
run fused1 X=1
run split1 WOQ_ATTN_O=0
run fused2 X=1
run split2 WOQ_ATTN_O=0
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0 --steps 32 --warmup 8 > $OUT/bench_prof.json 2> $OUT/rocprof.err
python tools/prof_stats.py $(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1) 6 > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt; rm -rf $OUT/prof
