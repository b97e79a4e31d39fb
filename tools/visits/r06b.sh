#!/bin/bash
# r06b: stage stamps of the product decode GEMV (woq_gemv_xqs.h) and of the long-lived-workgroup kernel v2
# (woq_gemv_xqm.h, coalesced prologue), then the quick bench line per geometry knob
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
for sh in "qkv 8 4" "o 8 4" "gate_up 4 4" "down 8 4"; do
  set -- $sh
  timeout 120 tools/xq_probe_stamps.bin 3 $1 $2 $3 > $O/stamps_$1.txt 2>&1
  WOQ_XQM_WG_PER_CU=2 timeout 120 tools/xq_probe_stamps.bin 1 $1 $2 $3 2>&1 | grep -A8 "xqm timeline" > $O/stamps_$1_wg2.txt
  grep -B1 -A8 "timeline" $O/stamps_$1.txt; cat $O/stamps_$1_wg2.txt
done
timeout 300 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_xqm.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_xqm.txt | cut -c1-220
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
run xqm0 WOQ_XQM=0
run xqm1 WOQ_XQM=1
run xqm1_wg2 WOQ_XQM=1 WOQ_XQM_WG_PER_CU=2
run xqm1_short8 WOQ_XQM=1 WOQ_XQM_SHORTK=8
run xqm1_short8_wg2 WOQ_XQM=1 WOQ_XQM_SHORTK=8 WOQ_XQM_WG_PER_CU=2
run xqm1_wg3 WOQ_XQM=1 WOQ_XQM_WG_PER_CU=3
run xqm1_long12 WOQ_XQM=1 WOQ_XQM_LONGK=12
run xqm0_b WOQ_XQM=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06b/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-22s tok/s %7.1f 128: %7.1f frac %.4f us %.3f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), r["frac"], r["us_per_launch"]),
              {k: v["us"] for k, v in r["by_projection"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
