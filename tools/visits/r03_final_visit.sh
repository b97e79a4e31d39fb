#!/bin/bash
# full measurement visit: the driver's bench command, its rocprofv3 kernel trace, the PMC traffic pass
set -u
TAG=${1:-r03ar}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity > $OUT/bench_prof.json 2> $OUT/rocprof.err; echo "rocprof rc=$?"
DB=$(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1)
python tools/prof_stats.py $DB 14 > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
python tools/prof_timeline.py $DB > $OUT/token_timeline.txt 2>&1; head -16 $OUT/token_timeline.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 > $OUT/pmc_bench.json 2> $OUT/pmc.err; echo "pmc rc=$?"
python tools/pmc_traffic.py $OUT/pmc $OUT/pmc_traffic.json gemv_xqs > $OUT/pmc_summary.txt 2>&1; tail -12 $OUT/pmc_summary.txt
rm -rf $OUT/pmc $OUT/prof
