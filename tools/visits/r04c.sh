#!/bin/bash
# r04c: merge in one round trip, raw fp8 registers (two workgroups per CU), tests, long-context A/B, o_proj wave geometry A/B, full bench line
set -u
TAG=r04c; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_attention_fullgeom.py "tests/test_gpu_fullsize_oracle.py::test_in_launch_handoffs_equal_separate_launches" -q --durations=8 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -16 $OUT/pytest.log
timeout 600 python tools/longctx_ab.py 16 8192 fp8 > $OUT/longctx_ab.txt 2>&1; cat $OUT/longctx_ab.txt
DEC="--no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures --steps 128 --warmup 16"
timeout 300 python bench.py $DEC > $OUT/dec128.json 2> $OUT/dec128.err
WOQ_XQ_TPW_SHORT=4 timeout 300 python bench.py $DEC > $OUT/dec128_tpw4.json 2> $OUT/dec128_tpw4.err
timeout 300 python bench.py $DEC > $OUT/dec128_b.json 2> $OUT/dec128_b.err
for f in dec128 dec128_tpw4 dec128_b; do python -c "
import json
d=json.load(open('$OUT/$f.json')); print('$f', round(d['value'],1), {k:round(v['us_per_launch'],2) for k,v in d['roofline'].get('by_projection',{}).items() if isinstance(v,dict)})"; done
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "cond", d.get("clock_conditioning"))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac")})
print("prefill", {k:v for k,v in d["prefill"].items() if k in ("tokens_per_s","mfma_frac","tflops")})
print("parity.prefill.attention", d["parity"]["prefill"].get("attention"))
for e in d["extra_configs"]:
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ("config","decode_tokens_per_s","ms_per_token","hbm_frac_weights","hbm_frac_weights_plus_kv","tokens_per_s","mfma_frac","workload")})
PY
