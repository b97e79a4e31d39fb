#!/bin/bash
# r06am: down projection (86 K tiles) as 15 waves of 6 tiles instead of 11 waves of 8 (A/B build -DWOQ_XQ_TPW6, WOQ_XQ_TPW_LONG=6)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06am; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
WOQ_HIP_LIB=$PWD/tools/lib_xq_t6.so WOQ_XQ_TPW_LONG=6 timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "vs_oracle or graph" > $O/pytest_t6.txt 2>&1; echo "pytest t6 rc=$?"; tail -2 $O/pytest_t6.txt | cut -c1-200
for rep in 1 2 3; do
  run def_$rep X=1
  run t6off_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_t6.so
  run t6_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_t6.so WOQ_XQ_TPW_LONG=6
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06am/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        bp = d["roofline"]["by_projection"]
        print("%-12s tok/s %7.1f 128: %7.1f  frac %.4f  qkv %.2f o %.2f gate_up %.2f down %.2f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), d["roofline"]["frac"], bp["qkv"]["us"], bp["o"]["us"], bp["gate_up"]["us"], bp["down"]["us"]))
    except Exception as e:
        print(f, "ERR", e)
PY
