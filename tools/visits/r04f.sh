#!/bin/bash
# r04f: is the 0.77 / 0.87 ms bimodality of the 16-layer Mistral 8k decode a property of the engine instance (memory placement)?
set -u
TAG=r04f; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/longctx_ab.py 16 8192 fp8 0:0:32 0:0:32 0:0:32 0:0:32 0:0:32 0:0:32 0:0:32 0:0:32 > $OUT/repeat_one_process.txt 2>&1; grep fold $OUT/repeat_one_process.txt
for i in 1 2 3 4; do timeout 300 python tools/longctx_ab.py 16 8192 fp8 0:0:32 2>&1 | grep fold; done | tee $OUT/repeat_processes.txt
DEC="--no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures --steps 128 --warmup 16"
for i in 1 2 3 4 5; do timeout 300 python bench.py $DEC 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('7B dec128', round(d['value'],1))"; done | tee $OUT/repeat_7b.txt
