#!/bin/bash
# r06n: fused qkv + attention launch — attention over the cache on q alone, new position as a fifth partial; q strips
# with the deeper window. Parity of the attention paths, then same-box A/B of the window depths against the r05 library.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_attention_fullgeom.py tests/test_gpu_fullsize_oracle.py -q -m gpu --maxfail=10 -k "not prefill_gemm" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt | cut -c1-220
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
for rep in 1 2; do
  run d83_$rep X=1
  run r05_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_r05.so
  run d44_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d44.so
  run d84_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d84.so
  run d82_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d82.so
  run d64_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_d64.so
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06n/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s tok/s %7.1f 128: %7.1f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0)))
    except Exception as e:
        print(f, "ERR", e)
PY
