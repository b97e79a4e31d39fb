#!/bin/bash
# r05zz: the round's LAST tree (graph replays of 8 chained steps, workspace change): full GPU suite, smoke, the driver's
# bench command, the default bench, kernel trace of the act-order engine
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05zz; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o ao -- python tools/visits/r05d.py > $O/act_order_prof.txt 2> $O/rocprof.err; echo "rocprof rc=$?"
python tools/prof_stats.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) 12 > $O/kernel_stats_act_order.txt 2>&1; cut -c1-170 $O/kernel_stats_act_order.txt; rm -rf $O/prof
python - <<'PY'
import json
for f in ("bench20", "bench"):
    try:
        d = json.loads(open(f"gpurun_out/r05zz/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "128-step", round(d.get("value_128_steps", 0), 1), "roofline", round(d["roofline"]["frac"], 4),
              "traffic", d["roofline"]["traffic"], "prefill", round(d.get("prefill", {}).get("mfma_frac", 0), 4),
              {k: round(v, 1) for k, v in d["launch_modes"].items() if isinstance(v, float)})
        print("   ", d.get("configs_summary"))
    except Exception as e:
        print(f, "ERR", e)
PY
