#!/bin/bash
# r06z: e4m3 cache — rings of four K passes / four V runs in the per-head attention; waves_per_eu(4) on the fused kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attention_fullgeom.py tests/test_gpu_engine.py -m gpu -x -q -k "fp8 or fused_launch or sliced or slices or window or mistral or e4m3" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:16:1:0 0:0:24:2:1 0:0:12:1:0 0:0:16:1:0 > $O/ab_mistral_8k.txt 2> $O/ab_mistral_8k.err
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 4096 fp8 0:0:16:1:0 0:0:8:1:0 > $O/ab_mistral_4k.txt 2> $O/ab_mistral_4k.err
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 2048 fp8 0:0:16:1:0 0:0:8:1:0 > $O/ab_mistral_2k.txt 2> $O/ab_mistral_2k.err
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 128 fp8 0:0:1:1:0 > $O/ab_mistral_128.txt 2> $O/ab_mistral_128.err
LCAB_GRAPH=1 timeout 900 python tools/longctx_ab.py 16 16384 fp8 0:0:16:1:0 0:0:32:2:1 > $O/ab_mistral_16k.txt 2> $O/ab_mistral_16k.err
python tools/ab_print.py $O/ab_*.txt
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
for rep in 1 2; do timeout 200 python bench.py $Q > $O/q_$rep.json 2> $O/q_$rep.err; python -c "
import json;d=json.loads(open('$O/q_$rep.json').read().strip().splitlines()[-1]);print('headline',d['value'],d.get('value_128_steps'))"; done
