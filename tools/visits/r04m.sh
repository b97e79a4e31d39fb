#!/bin/bash
# r04m: four-set rolling K / V windows in the decode attention: tests, decode line (128 / 512 steps), threshold sweep again
set -u
TAG=r04m; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_attention_fullgeom.py "tests/test_gpu_fullsize_oracle.py::test_in_launch_handoffs_equal_separate_launches" -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
DEC="--no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures --condition-ms 300"
for st in "20 5" "128 16" "512 32"; do set -- $st
  timeout 300 python bench.py --steps $1 --warmup $2 $DEC 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('steps $1:', round(d['value'],1), round(d['launch_modes']['hipgraph_replay_tokens_per_s'],1))"
done | tee $OUT/decode.txt
for ctx in 160 256 384 512 640 768; do
  LCAB_KVH=32 LCAB_INTER=11008 timeout 300 python tools/longctx_ab.py 16 $ctx fp16 0:0:1 0:0:8 2>/dev/null | grep fold | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print('ctx $ctx splits', d['splits'], d['ms_per_token'])"
done | tee $OUT/threshold_sweep.txt
