#!/bin/bash
# r03g: where does the decode token go? graph replay with launches left out (WOQ_ENGINE_SKIP), plus rocprof kernel stats
set -u
OUT=gpurun_out/r03g
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --no-extra --no-cpu-baseline --no-parity --prefill-seqs 0 --steps 128 --warmup 16"
for m in 0 1 2 4 8 16 32 3 31 63 29; do
  v=$(WOQ_ENGINE_SKIP=$m timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1000)")
  echo "skip mask $m : $v us per token" | tee -a $OUT/skip_masks.txt
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- $B --steps 32 --warmup 8 > $OUT/bench_prof.json 2> $OUT/rocprof.err
python tools/prof_stats.py $(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1) 12 > $OUT/kernel_stats.txt 2>&1
cat $OUT/kernel_stats.txt
