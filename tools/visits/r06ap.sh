#!/bin/bash
# r06ap: polling interval of the attention workgroups' granule waits (s_sleep 0 / 2 / 8); window depth 8 on the 8-tile waves re-checked after the request-order change
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ap; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
for rep in 1 2 3; do
  run def_$rep X=1
  run ps0_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_ps0.so
  run ps8_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_ps8.so
  run d8_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d8.so
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06ap/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        bp = d["roofline"]["by_projection"]
        print("%-12s tok/s %7.1f 128: %7.1f  frac %.4f  qkv %.2f o %.2f gate_up %.2f down %.2f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), d["roofline"]["frac"], bp["qkv"]["us"], bp["o"]["us"], bp["gate_up"]["us"], bp["down"]["us"]))
    except Exception as e:
        print(f, "ERR", e)
PY
