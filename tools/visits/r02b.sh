#!/bin/bash
# r02b: where does the decode GEMV's time go (knock-outs, timeline), and the TPW=4 geometry A/B
set -u
OUT=gpurun_out/r02b; mkdir -p $OUT
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | head -20 >> $OUT/host.txt; free -g >> $OUT/host.txt
timeout 300 tools/gemv_probe.bin 5 > $OUT/probe.txt 2>&1; echo "probe rc=$?"
timeout 300 tools/gemv_probe_stamps.bin 2 > $OUT/probe_stamps.txt 2>&1; echo "stamps rc=$?"
B="python bench.py --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
for i in 1 2; do
timeout 300 $B > $OUT/bench_base_$i.json 2>$OUT/err.txt; echo base $(python -c "import json;d=json.load(open('$OUT/bench_base_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
WOQ_TILE_TPW4=32 timeout 300 $B > $OUT/bench_tpw4_$i.json 2>$OUT/err.txt; echo tpw4 $(python -c "import json;d=json.load(open('$OUT/bench_tpw4_$i.json'));print(d['value'], d['roofline']['us_per_launch'])")
done
cat $OUT/probe.txt
