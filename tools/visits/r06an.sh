#!/bin/bash
# r06an: e4m3 cache — one shared ring of K + V register sets (8 K passes + 2 V runs in flight before q, K sets re-used for V runs), buffer-descriptor requests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06an; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attention_fullgeom.py tests/test_gpu_engine.py -m gpu -x -q -k "fp8 or fused_launch or sliced or slices or window or mistral or e4m3" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-200
lc() { name=$1; shift; args=$1; shift; env "$@" timeout 300 python tools/longctx_bench.py $args > $O/lc_$name.json 2> $O/lc_$name.err; echo "$name rc=$? $(python -c "
import json,sys
d=json.loads(open('$O/lc_$name.json').read().strip().splitlines()[-1]); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('ctx','splits','decode_ms','decode_tok_s','grouped')})" 2>&1 | tail -1)"; }
for rep in 1 2; do
  for ctx in 8192 2048 128 16384; do
    lc ring82_${ctx}_$rep "$ctx 2048 fp8" X=1
    lc ring44_${ctx}_$rep "$ctx 2048 fp8" WOQ_HIP_LIB=$PWD/tools/lib_xq_ring44.so
    lc ring80_${ctx}_$rep "$ctx 2048 fp8" WOQ_HIP_LIB=$PWD/tools/lib_xq_ring80.so
  done
done
