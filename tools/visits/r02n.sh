#!/bin/bash
set -u
OUT=gpurun_out/r02n; mkdir -p $OUT
python - <<'PY' > $OUT/tp1_ab.txt 2>&1
import os, sys, json, subprocess
for tag, env in (("xq", {}), ("f32", {"WOQ_ENGINE_XQ": "0"}), ("xq2", {}), ("f32b", {"WOQ_ENGINE_XQ": "0"})):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "70b", "--steps", "32", "--warmup", "4", "--no-extra"], env=e, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(tag, round(d["value"], 2), round(d["roofline"]["us_per_launch"], 3), d["roofline"]["kernel"][:24])
    except Exception as ex:
        print(tag, "failed", r.stderr[-400:])
PY
cat $OUT/tp1_ab.txt
