#!/bin/bash
# r04a: new attention parity tests; baseline decode line; PMC passes over the decode GEMVs; profiled-vs-unprofiled gap
set -u
TAG=r04a; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention_fullgeom.py "tests/test_gpu_api.py::test_generate_reads_the_engine_status_and_retries_or_raises" -q -s --durations=10 > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest_new.log
DEC="--no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures"
timeout 300 python bench.py --steps 128 --warmup 16 $DEC > $OUT/dec128.json 2> $OUT/dec128.err; echo "dec128 rc=$?"
python -c "import json;d=json.load(open('$OUT/dec128.json'));print('dec128', d['value'], d['roofline'].get('frac'), d['roofline'].get('by_projection'))"
timeout 300 python bench.py --steps 20 --warmup 5 $DEC > $OUT/dec20.json 2> $OUT/dec20.err
python -c "import json;d=json.load(open('$OUT/dec20.json'));print('dec20', d['value'])"
# (e) the gap: same command with and without the profiler, clocks sampled from sysfs
python tools/clock_sampler.py 40 0.01 > $OUT/clk_plain.jsonl &
CS=$!
timeout 300 python bench.py --steps 512 --warmup 32 $DEC > $OUT/gap_plain.json 2> $OUT/gap_plain.err
kill $CS 2>/dev/null; wait $CS 2>/dev/null
python tools/clock_sampler.py 60 0.01 > $OUT/clk_prof.jsonl &
CS=$!
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_gap -o bench -- python bench.py --steps 512 --warmup 32 $DEC > $OUT/gap_prof.json 2> $OUT/gap_prof.err
kill $CS 2>/dev/null; wait $CS 2>/dev/null
for k in plain prof; do
  python - <<PY
import json,subprocess
d=json.load(open("$OUT/gap_$k.json"))
t0,t1=d["timed_region_unix"]
print("$k", "tok/s", d["value"], "ms", d["ms_per_step"])
subprocess.run(["python","tools/clock_sampler.py","--summarise","$OUT/clk_$k.jsonl",str(t0),str(t1)])
PY
done
DB=$(ls $OUT/prof_gap/*.db $OUT/prof_gap/*/*.db 2>/dev/null | head -1)
python tools/prof_stats.py $DB 12 > $OUT/gap_kernel_stats.txt 2>&1; head -20 $OUT/gap_kernel_stats.txt
rm -rf $OUT/prof_gap
# PMC passes over the decode step (graph replays): where the GEMV's wave cycles go
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1
grep -c "" $OUT/list_avail.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES"
P3="SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAVE_CYCLES GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -- python bench.py --steps 8 --warmup 2 $DEC > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc$i rc=$?"; tail -2 $OUT/pmc$i.err | cut -c1-200
  python tools/pmc_summary.py $OUT/pmc$i gemv_xqs >> $OUT/pmc_gemv.txt 2>&1
  python tools/pmc_summary.py $OUT/pmc$i lm_head >> $OUT/pmc_gemv.txt 2>&1
  rm -rf $OUT/pmc$i
done
cat $OUT/pmc_gemv.txt | cut -c1-150
