#!/bin/bash
# r06t: slices of a head merge among themselves (tagged partial granules, no combine launch) — parity, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_attention_fullgeom.py -m gpu -x -q -k "fused_launch or sliced_regime or 4x2048" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 2048 fp16 0:0:16:0 0:0:8:1 0:0:4:1 0:0:16:1 0:0:12:1 0:0:8:1 > $O/ab_7b_2048.txt 2> $O/ab_7b_2048.err
echo "7b rc=$?"; python tools/ab_print.py $O/ab_7b_2048.txt
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 512 fp16 0:0:8:0 0:0:4:1 0:0:2:1 0:0:8:1 0:0:1:1 > $O/ab_7b_512.txt 2> $O/ab_7b_512.err
echo "7b512 rc=$?"; python tools/ab_print.py $O/ab_7b_512.txt
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 320 fp16 0:0:1:1 0:0:4:1 0:0:2:1 > $O/ab_7b_320.txt 2> $O/ab_7b_320.err
echo "7b320 rc=$?"; python tools/ab_print.py $O/ab_7b_320.txt
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:0:0 0:0:16:1:0 0:0:12:1:0 > $O/ab_mistral_8k.txt 2> $O/ab_mistral_8k.err
echo "mistral rc=$?"; python tools/ab_print.py $O/ab_mistral_8k.txt
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 4096 fp8 0:0:16:0:1 0:0:8:1:0 0:0:16:1:0 > $O/ab_mistral_4k.txt 2> $O/ab_mistral_4k.err
echo "mistral4k rc=$?"; python tools/ab_print.py $O/ab_mistral_4k.txt
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 2048 fp8 0:0:8:0:0 0:0:8:1:0 0:0:4:1:0 0:0:16:1:0 > $O/ab_mistral_2k.txt 2> $O/ab_mistral_2k.err
echo "mistral2k rc=$?"; python tools/ab_print.py $O/ab_mistral_2k.txt
