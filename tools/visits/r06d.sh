#!/bin/bash
# r06d: the decode GEMV with its whole window in front of the argument-segment reads (XqsLate) — parity, stage stamps,
# same-box A/B against the round-5 library, the driver's bench command with the compact line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_tp_device.py -q -m gpu --maxfail=10 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-200
WOQ_HIP_LIB=$PWD/tools/lib_xq_stamps.so timeout 200 python tools/xqs_stamps.py > $O/stamps.txt 2> $O/stamps.err; echo "stamps rc=$?"; cat $O/stamps.txt
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
run new_1 X=1
run r05_1 WOQ_HIP_LIB=$PWD/tools/lib_xq_r05.so
run new_2 X=1
run r05_2 WOQ_HIP_LIB=$PWD/tools/lib_xq_r05.so
run new_3 X=1
run r05_3 WOQ_HIP_LIB=$PWD/tools/lib_xq_r05.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06d/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-14s tok/s %7.1f 128: %7.1f frac %.4f us %.3f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), r["frac"], r["us_per_launch"]),
              {k: v["us"] for k, v in r["by_projection"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"; wc -c $O/bench20.json; cat $O/bench20.json; cp bench_extra.json $O/bench20_extra.json
