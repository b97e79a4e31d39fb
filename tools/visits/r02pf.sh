#!/bin/bash
# r02pf: A/B of the attention launch's prefetch rider (WOQ_ENGINE_PF_BLOCKS / _MB) on the driver's decode bench
mkdir -p gpurun_out
for cfg in "0 0" "224 0" "224 16" "224 32" "448 16"; do
  set -- $cfg
  r=$(WOQ_ENGINE_PF_BLOCKS=$1 WOQ_ENGINE_PF_MB=$2 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --steps 128 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f tok/s  %.4f ms' % (d['value'], d['ms_per_step']))")
  echo "blocks $1 gate/up MB $2: $r"
done 2>&1 | tee gpurun_out/r02pf.txt
