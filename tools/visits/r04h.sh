#!/bin/bash
# r04h: eager bursts as the default decode launch mode: tests, decode line, full bench line, kernel trace
set -u
TAG=r04h; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py tests/test_gpu_tp_device.py -q --durations=5 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest.log
DEC="--no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures"
for st in "20 5" "128 16" "512 32"; do set -- $st
  timeout 300 python bench.py --steps $1 --warmup $2 $DEC > $OUT/dec$1.json 2> $OUT/dec$1.err
  python -c "
import json
a=json.load(open('$OUT/dec$1.json'))
print('steps $1:', round(a['value'],1), a['launch_modes']['eager_bursts_tokens_per_s'], a['launch_modes']['hipgraph_replay_tokens_per_s'], 'roofline', round(a['roofline']['frac'],4), {k:round(v['us_per_launch'],2) for k,v in a['roofline'].get('by_projection',{}).items() if isinstance(v,dict)}, 'ceiling', {k:round(v,3) for k,v in a['roofline'].get('ceiling',{}).items() if isinstance(v,float)})"
done
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["launch_modes"])
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","us_per_launch")})
print("structures", d.get("other_launch_structures"))
print("prefill", {k:v for k,v in d["prefill"].items() if k in ("tokens_per_s","mfma_frac")})
for e in d["extra_configs"]:
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ("config","decode_tokens_per_s","ms_per_token","hbm_frac_weights","hbm_frac_weights_plus_kv","tokens_per_s","mfma_frac","workload")})
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity > $OUT/bench_prof.json 2> $OUT/rocprof.err; echo "rocprof rc=$?"
DB=$(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1)
python tools/prof_stats.py $DB 14 > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt | cut -c1-70,92-170
python tools/prof_timeline.py $DB > $OUT/token_timeline.txt 2>&1; head -12 $OUT/token_timeline.txt
rm -rf $OUT/prof
