#!/bin/bash
# r04zz: the round's last tree: full GPU suite, smoke, the driver's two bench commands
set -u
TAG=r04zz; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench(20) rc=$?"
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench_steps20", "bench"):
    d=json.load(open("$OUT/%s.json" % f))
    print(f, round(d["value"],1), {k:round(v,1) for k,v in d["launch_modes"].items() if isinstance(v,float)}, round(d["roofline"]["frac"],4), round(d["prefill"]["mfma_frac"],4), [round(e.get("decode_tokens_per_s", e.get("tokens_per_s", 0)),1) for e in d["extra_configs"]])
PY
