#!/bin/bash
# r04k: rolling-window depth A/B (4 / 6 / 8 tiles in flight per wave), fused-vs-sliced attention threshold sweep
set -u
TAG=r04k; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
DEC="--no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures --steps 128 --warmup 16 --condition-ms 300"
for rep in 1 2; do
for v in base d6 d8; do
  if [ $v = base ]; then unset WOQ_HIP_LIB; else export WOQ_HIP_LIB=$PWD/tools/lib_xq_$v.so; fi
  timeout 300 python bench.py $DEC 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$v rep$rep', round(d['value'],1), {k:round(x['us_per_launch'],2) for k,x in d['roofline'].get('by_projection',{}).items() if isinstance(x,dict)}, 'family', round(d['roofline']['us_per_launch'],3))"
done; done | tee $OUT/depth_ab.txt
unset WOQ_HIP_LIB
for ctx in 200 256 320 384 512; do
  LCAB_KVH=32 LCAB_INTER=11008 timeout 300 python tools/longctx_ab.py 16 $ctx fp16 0:0:1 0:0:0 2>/dev/null | grep fold | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print('ctx $ctx splits', d['splits'], d['ms_per_token'])"
done | tee $OUT/threshold_sweep.txt
