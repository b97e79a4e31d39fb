#!/bin/bash
# graph-on-current-stream tree: engine / api / tp tests, the driver's bench command, the default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ac
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py tests/test_gpu_tp_device.py -x -q -m gpu > gpurun_out/r04ac/pytest.txt 2>&1
tail -3 gpurun_out/r04ac/pytest.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04ac/bench20.json 2> gpurun_out/r04ac/bench20.err
timeout 400 python bench.py > gpurun_out/r04ac/bench.json 2> gpurun_out/r04ac/bench.err
python - <<'PY'
import json
for f in ("bench20","bench"):
    try:
        d=json.loads(open(f"gpurun_out/r04ac/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("launch_modes"), d["roofline"]["frac"], d["roofline"].get("avg_launch_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
