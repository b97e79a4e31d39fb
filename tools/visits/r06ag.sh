#!/bin/bash
# r06ag: wave geometry of the gate/up pairs (8 waves x 4 tiles, default, vs 4 waves x 8 tiles: WOQ_XQ_TPW=8) and of o_proj
# (4 x 8 default vs 8 x 4: WOQ_XQ_TPW_SHORT=4) on the round-6 kernel; environment switches of csrc/woq_gemv_xq.hip
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ag; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
for rep in 1 2 3; do
  run def_$rep X=1
  run gu8_$rep WOQ_XQ_TPW=8
  run o4_$rep WOQ_XQ_TPW_SHORT=4
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06ag/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        bp = d["roofline"]["by_projection"]
        print("%-12s tok/s %7.1f 128: %7.1f  frac %.4f  qkv %.2f o %.2f gate_up %.2f down %.2f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), d["roofline"]["frac"], bp["qkv"]["us"], bp["o"]["us"], bp["gate_up"]["us"], bp["down"]["us"]))
    except Exception as e:
        print(f, "ERR", e)
PY
