#!/bin/bash
# r06y: packed fp8 -> fp32 conversions in the per-head decode attention; full GPU suite on the tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
export TMPDIR=/tmp
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:16:1:0 0:0:24:2:1 0:0:12:1:0 0:0:16:1:0 > $O/ab_mistral_8k.txt 2> $O/ab_mistral_8k.err
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 4096 fp8 0:0:16:1:0 0:0:8:1:0 > $O/ab_mistral_4k.txt 2> $O/ab_mistral_4k.err
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 2048 fp8 0:0:16:1:0 0:0:8:1:0 > $O/ab_mistral_2k.txt 2> $O/ab_mistral_2k.err
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 128 fp8 0:0:1:1:0 > $O/ab_mistral_128.txt 2> $O/ab_mistral_128.err
python tools/ab_print.py $O/ab_*.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
