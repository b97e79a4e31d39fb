#!/bin/bash
# r05m: q's RoPE inside the prefill attention kernel (RoPE / append pass for k and v only) vs the in-place pass: parity, timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_attention_fullgeom.py tests/test_gpu_engine.py -q -m gpu -k "prefill or attention or fullgeom or chunk or mistral or window or fp8_kv" 2>&1 | grep -E "passed|failed|error" | tail -3
for mode in attn pass attn pass; do
  WOQ_PREFILL_ROPE_Q=$mode timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-parity --no-structures > $O/b_$mode.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_$mode.json").read().strip().splitlines()[-1])
print("$mode", "prefill mfma_frac", round(d["prefill"]["mfma_frac"],4), "ms", round(d["prefill"]["ms"],2))
PY
done
