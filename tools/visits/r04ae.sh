#!/bin/bash
# last-tree records of round 4: table-type tests + timings (incl. the perm-mask variant), then the full GPU suite,
# smoke, the driver's bench command and the default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ae; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py tests/test_gpu_api.py -q -m gpu -k "table" > $O/pytest_table.txt 2>&1
tail -2 $O/pytest_table.txt
timeout 200 python tools/table_decode_bench.py --engine > $O/table_bench.txt 2> $O/table_bench.err
cat $O/table_bench.txt
WOQ_HIP_LIB=$PWD/tools/lib_xq_lutperm.so timeout 200 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "engine_table" > $O/pytest_lutperm.txt 2>&1
tail -2 $O/pytest_lutperm.txt
WOQ_HIP_LIB=$PWD/tools/lib_xq_lutperm.so timeout 120 python tools/table_decode_bench.py --engine --types nf4,nf4:bf16 > $O/table_bench_lutperm.txt 2> $O/table_bench_lutperm.err
grep engine $O/table_bench_lutperm.txt
timeout 700 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -1 $O/smoke.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench20","bench"):
    try:
        d=json.loads(open(f"gpurun_out/r04ae/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],1), {k:round(v,1) for k,v in d["launch_modes"].items() if isinstance(v,float)}, round(d["roofline"]["frac"],4), round(d["prefill"]["mfma_frac"],4))
        for e in d["extra_configs"]:
            print("   ", e.get("config","?")[:90], round(e.get("decode_tokens_per_s", e.get("tokens_per_s", 0)),1))
    except Exception as e:
        print(f, "ERR", e)
PY
