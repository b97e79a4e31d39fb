#!/bin/bash
# kernel trace of the last tree's bench command (decode + roofline + 4 x 2048 prompt pass; extras, CPU leg, parity legs
# and the persistent-launch structure left out to fit the remaining GPU time)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ag; mkdir -p $O
export TMPDIR=/tmp
S=$(date +%s)
timeout 100 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity --no-structures > $O/bench_prof.json 2> $O/rocprof.err
echo "rocprof rc=$? in $(( $(date +%s) - S )) s"
DB=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1)
python tools/prof_stats.py $DB 16 > $O/kernel_stats.txt 2>&1; cut -c1-70,92-170 $O/kernel_stats.txt
rm -rf $O/prof
python -c "
import json
d=json.loads(open('$O/bench_prof.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'], d['launch_modes'])
"
