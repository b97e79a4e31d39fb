#!/bin/bash
# r06x: o_proj strips as the fused launch's third role (tagged XQ granules) — parity, then headline A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attention_fullgeom.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_engine.py -m gpu -x -q -k "fused or handoff or launches or grouped_attention_long or self_merge or merging" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.log
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
for rep in 1 2 3; do
  run o1_$rep WOQ_FUSE_OPROJ=1
  run o0_$rep WOQ_FUSE_OPROJ=0
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06x/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s tok/s %7.1f 128: %7.1f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0)))
    except Exception as e:
        print(f, "ERR", e)
PY
