#!/bin/bash
# r04b: attention tests (full geometry, fixed-chunk slices, in-launch merge), decode line with clock conditioning, long-context A/B, full bench line
set -u
TAG=r04b; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_attention_fullgeom.py "tests/test_gpu_fullsize_oracle.py::test_in_launch_handoffs_equal_separate_launches" "tests/test_gpu_fullsize_oracle.py::test_llama2_7b_shape_decoder_logits_vs_oracle" -q -x --durations=8 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -22 $OUT/pytest.log
DEC="--no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures"
for st in "20 5" "128 16"; do set -- $st
  timeout 300 python bench.py --steps $1 --warmup $2 $DEC > $OUT/dec$1.json 2> $OUT/dec$1.err
  timeout 300 python bench.py --steps $1 --warmup $2 $DEC --condition-ms 0 > $OUT/dec$1_nocond.json 2> $OUT/dec$1_nocond.err
  python -c "
import json
a=json.load(open('$OUT/dec$1.json')); b=json.load(open('$OUT/dec$1_nocond.json'))
print('steps $1: conditioned', round(a['value'],1), a['clock_conditioning'], '| unconditioned', round(b['value'],1), '| roofline frac', a['roofline']['frac'])"
done
timeout 600 python tools/longctx_ab.py 16 8192 fp8 > $OUT/longctx_ab.txt 2>&1; cat $OUT/longctx_ab.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "cond", d.get("clock_conditioning"))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac")}, "by_proj", {k:round(v["frac"],3) for k,v in d["roofline"].get("by_projection",{}).items() if isinstance(v,dict)})
print("prefill", d["prefill"].get("tokens_per_s"), d["prefill"].get("mfma_frac"))
print("parity.prefill.attention", d["parity"]["prefill"].get("attention"))
for e in d["extra_configs"]:
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ("config","decode_tokens_per_s","ms_per_token","hbm_frac_weights","hbm_frac_weights_plus_kv","tokens_per_s","mfma_frac","workload")})
PY
