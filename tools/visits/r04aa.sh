#!/bin/bash
# r04aa: fused qkv + attention launch for grouped-query / windowed shapes: parity tests, Mistral-7B short-context decode A/B
set -u
TAG=r04aa; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attention_fullgeom.py tests/test_gpu_engine.py tests/test_gpu_api.py -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for ctx in 64 256; do
for fuse in 1 0; do
  WOQ_ENGINE_FUSE_ATTN=$fuse timeout 300 python tools/longctx_ab.py 32 $ctx fp16 0:0:1 2>/dev/null | grep fold | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print('mistral-7b shape 32 layers ctx $ctx fuse_attn=$fuse', d['ms_per_token'], 'ms/token', round(1000/d['ms_per_token'],1), 'tokens/s')"
done; done | tee $OUT/mistral_short_ctx.txt
