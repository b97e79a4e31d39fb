#!/bin/bash
# r06af: window depth of the stand-alone decode GEMVs (o / gate-up / down) 4 (default) vs 6 vs 8 on the round-6 kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06af; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
for rep in 1 2 3; do
  run d7_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d7.so; run d4_$rep X=1
  run d5_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d5.so
  run d7_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d7.so; run d4_$rep X=1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06af/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        bp = d["roofline"]["by_projection"]
        print("%-12s tok/s %7.1f 128: %7.1f  frac %.4f  qkv %.2f o %.2f gate_up %.2f down %.2f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), d["roofline"]["frac"], bp["qkv"]["us"], bp["o"]["us"], bp["gate_up"]["us"], bp["down"]["us"]))
    except Exception as e:
        print(f, "ERR", e)
PY
