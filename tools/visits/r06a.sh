#!/bin/bash
# r06a: first visit of round 6 — the long-lived-workgroup decode GEMV (woq_gemv_xqm.h) against the one-workgroup-per-strip
# kernel: parity (engine / parity / full-size suites), then same-box A/B of the quick bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -q -m gpu --maxfail=15 > $O/pytest_xqm.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_xqm.txt | cut -c1-220
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --no-structures --prefill-seqs 0"
for rep in 1 2; do
  for v in 0 1; do
    WOQ_XQM=$v timeout 200 python bench.py $Q > $O/q_xqm${v}_$rep.json 2> $O/q_xqm${v}_$rep.err; echo "xqm=$v rep $rep rc=$?"
    cp bench_extra.json $O/q_xqm${v}_${rep}_extra.json 2>/dev/null
  done
done
WOQ_XQM=1 WOQ_XQM_LONGK=12 timeout 200 python bench.py $Q > $O/q_longk12.json 2> $O/q_longk12.err
WOQ_XQM=1 WOQ_ENGINE_FUSE_ATTN=0 timeout 200 python bench.py $Q > $O/q_xqm1_nofuse.json 2> $O/q_nofuse.err
WOQ_XQM=0 WOQ_ENGINE_FUSE_ATTN=0 timeout 200 python bench.py $Q > $O/q_xqm0_nofuse.json 2> $O/q_nofuse0.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06a/q_*.json")):
    if f.endswith("_extra.json"): continue
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], "tok/s", round(d["value"], 1), "128:", round(d.get("value_128_steps", 0), 1), "frac", round(r["frac"], 4), "us", r["us_per_launch"],
              {k: v["us"] for k, v in r["by_projection"].items()}, r.get("ceiling"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:] if glob.glob(f.replace(".json", ".err")) else "")
PY
