#!/bin/bash
# r06c: stage stamps of the product decode GEMV inside the engine's launches and the A/B of the
# barrier-free last-arriver epilogue (item 2d)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
WOQ_HIP_LIB=$PWD/tools/lib_xq_stamps.so timeout 200 python tools/xqs_stamps.py > $O/stamps_base.txt 2> $O/stamps_base.err; echo "stamps rc=$?"; cat $O/stamps_base.txt
WOQ_HIP_LIB=$PWD/tools/lib_xq_last_stamps.so timeout 200 python tools/xqs_stamps.py > $O/stamps_last.txt 2> $O/stamps_last.err; echo "stamps last rc=$?"; cat $O/stamps_last.txt
WOQ_HIP_LIB=$PWD/tools/lib_xq_last.so timeout 300 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_last.txt 2>&1; echo "pytest(last) rc=$?"; tail -3 $O/pytest_last.txt | cut -c1-200
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
run base_1 X=1
run last_1 WOQ_HIP_LIB=$PWD/tools/lib_xq_last.so
run base_2 X=1
run last_2 WOQ_HIP_LIB=$PWD/tools/lib_xq_last.so
run short4 WOQ_XQ_TPW_SHORT=4
run last_short4 WOQ_HIP_LIB=$PWD/tools/lib_xq_last.so WOQ_XQ_TPW_SHORT=4
run base_3 X=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06c/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-22s tok/s %7.1f 128: %7.1f frac %.4f us %.3f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), r["frac"], r["us_per_launch"]),
              {k: v["us"] for k, v in r["by_projection"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
