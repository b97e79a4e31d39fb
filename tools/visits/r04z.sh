#!/bin/bash
# r04z: end-of-round records on the final tree: full GPU suite, smoke, the driver's bench commands, kernel trace, sampled-decode rate
set -u
TAG=r04z; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?"; tail -14 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
S=$(date +%s); timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"; tail -2 $OUT/bench.err
S=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver20.json 2> $OUT/bench_driver20.err; echo "bench(20) rc=$? in $(( $(date +%s) - S )) s"
python - <<PY
import json
for f in ("bench", "bench_driver20"):
    d=json.load(open("$OUT/%s.json" % f))
    print(f, "value", round(d["value"],1), "ms", round(d["ms_per_step"],4), {k:round(v,1) for k,v in d["launch_modes"].items() if isinstance(v,float)}, "e2e", round(d["hbm_frac_of_peak_end_to_end"],4), "roofline", round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"])
    print("  prefill", round(d["prefill"]["tokens_per_s"]), round(d["prefill"]["mfma_frac"],4), "gemm", round(d["prefill"].get("dominant_gemm",{}).get("kernel_mfma_frac",0),4), "parity", d["parity"]["decode"]["logits_max_abs_over_max_logit"], d["parity"]["prefill"]["worst_row_err_over_rowmax"], {k:v for k,v in d["parity"]["prefill"]["attention"]["per_sequence"].items()})
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "keys order tail", list(d.keys())[-5:])
    for e in d["extra_configs"]:
        print("  ", {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ("config","decode_tokens_per_s","hbm_frac_weights_plus_kv","tokens_per_s","mfma_frac")})
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity > $OUT/bench_prof.json 2> $OUT/rocprof.err; echo "rocprof rc=$?"
DB=$(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1)
python tools/prof_stats.py $DB 16 > $OUT/kernel_stats.txt 2>&1; cut -c1-70,92-170 $OUT/kernel_stats.txt
rm -rf $OUT/prof
timeout 600 python - > $OUT/sampled_rate.txt 2>&1 <<'PY'
import sys, time, json
import torch
sys.path.insert(0, ".")
from bench import LLAMA2_7B, build_engine
from intel_extension_for_transformers_amd.runtime.engine import DeviceSampler, generate_sampled
eng = build_engine(LLAMA2_7B, max_ctx=1024)
prompt = torch.randint(0, 32000, (32,)).tolist()
for name, smp in (("reference NeuralChat defaults (do_sample, T 0.1, top_k 40, top_p 0.75, penalty 1.1)", DeviceSampler(True, 0.1, 40, 0.75, 1.1)),
                  ("sampling without top_k (T 0.8, top_p 0.9): full-vocabulary sort", DeviceSampler(True, 0.8, 0, 0.9, 1.0)),
                  ("repetition penalty only (argmax)", DeviceSampler(False, 1.0, 0, 1.0, 1.1))):
    generate_sampled(eng, prompt, 16, smp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = generate_sampled(eng, prompt, 256, smp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"request": name, "new_tokens": len(out), "tokens_per_s_incl_prompt_pass": round(len(out) / dt, 1)}))
eng.generate(prompt, 16)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.generate(prompt, 256)
torch.cuda.synchronize()
print(json.dumps({"request": "engine.generate greedy (bursts of 16)", "tokens_per_s_incl_prompt_pass": round(256 / (time.perf_counter() - t0), 1)}))
PY
grep request $OUT/sampled_rate.txt
