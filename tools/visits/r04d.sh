#!/bin/bash
# r04d: long-context decode attention: forms A/B with the branch-free merge, slice-count sweep, per-kernel times
set -u
TAG=r04d; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_engine.py -q -k "slice or chunk or sliced" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python -m pytest "tests/test_gpu_api.py::test_plain_c_client_device_leg" -q > $OUT/pytest_c.log 2>&1; echo "pytest c rc=$?"; tail -3 $OUT/pytest_c.log
timeout 900 python tools/longctx_ab.py 16 8192 fp8 > $OUT/longctx_ab.txt 2>&1; cat $OUT/longctx_ab.txt
for v in "0:0:0" "1:0:0" "0:256:0"; do
  n=$(echo $v | tr ':' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$n -o lc -- python tools/longctx_ab.py 8 8192 fp8 $v > $OUT/prof_$n.txt 2>&1
  DB=$(ls $OUT/prof_$n/*.db $OUT/prof_$n/*/*.db 2>/dev/null | head -1)
  echo "== variant $v"; python tools/prof_stats.py $DB 9 2>&1 | cut -c1-60,92-170 | tee $OUT/kernels_$n.txt
  rm -rf $OUT/prof_$n
done
