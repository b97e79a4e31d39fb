#!/bin/bash
# r05z: last-tree records of round 5 — full GPU suite, smoke, the driver's bench command, the default bench, kernel trace
# and PMC traffic pass of the same tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity > $O/bench_prof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
python tools/prof_stats.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) 16 > $O/kernel_stats.txt 2>&1; cat $O/kernel_stats.txt | cut -c1-170; rm -rf $O/prof
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures --condition-ms 0 > $O/pmc_bench.json 2> $O/pmc.err; echo "pmc rc=$?"
python tools/pmc_traffic.py $O/pmc $O/pmc_traffic.json gemv_xqs > $O/pmc_summary.txt 2>&1; tail -4 $O/pmc_summary.txt | cut -c1-220; rm -rf $O/pmc
python - <<'PY'
import json
for f in ("bench20", "bench"):
    try:
        d = json.loads(open(f"gpurun_out/r05z/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "128-step", round(d.get("value_128_steps", 0), 1), "roofline", round(d["roofline"]["frac"], 4),
              "prefill", round(d.get("prefill", {}).get("mfma_frac", 0), 4))
        print("   ", d.get("configs_summary"))
    except Exception as e:
        print(f, "ERR", e)
PY
