#!/bin/bash
# r06ad: gate/up GEMV at six waves per SIMD (78 registers: the whole 688-workgroup grid resident in one round) vs the
# compiler's own 82 registers / five waves — same-box A/B of the headline, alternating; parity of the GEMV family
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ad; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_engine.py -m gpu -x -q -k "llama2_7b_shape or engine_logits or graph_replay or eager_bursts" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
for rep in 1 2 3; do
  run gu6_$rep X=1
  run gu5_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_gu5.so
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06ad/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        bp = d["roofline"]["by_projection"]
        print("%-12s tok/s %7.1f 128: %7.1f  frac %.4f  qkv %.2f o %.2f gate_up %.2f down %.2f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), d["roofline"]["frac"], bp["qkv"]["us"], bp["o"]["us"], bp["gate_up"]["us"], bp["down"]["us"]))
    except Exception as e:
        print(f, "ERR", e)
PY
