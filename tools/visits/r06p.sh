#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
for rep in 1 2 3; do
  run d83_$rep X=1
  run d81_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d81.so
  run d82_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d82.so
  run d62_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d62.so
  run r05_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_r05.so
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06p/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s tok/s %7.1f 128: %7.1f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0)))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py $Q > /dev/null 2> $O/rocprof.err
python tools/prof_stats.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) 6 2>&1 | cut -c1-170; rm -rf $O/prof
