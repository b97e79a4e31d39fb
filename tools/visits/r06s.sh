#!/bin/bash
# r06s: combine-led o_proj with ONE flag word per head (8 polled words per wave instead of 64) — parity, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_attention_fullgeom.py -m gpu -x -q -k "fused_launch_with or grouped_slices_with" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 2048 fp16 0:0:8:1 0:0:8:3 0:0:12:1 0:0:8:1 0:0:8:3 > $O/ab_7b_2048.txt 2> $O/ab_7b_2048.err
echo "7b rc=$?"; cat $O/ab_7b_2048.txt | cut -c1-200
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:0:0 0:0:0:2 0:0:0:0 0:0:0:2 > $O/ab_mistral_8k.txt 2> $O/ab_mistral_8k.err
echo "mistral rc=$?"; cat $O/ab_mistral_8k.txt | cut -c1-200
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 384 fp16 0:0:1:1 0:0:4:1 0:0:3:1 0:0:2:1 > $O/ab_7b_384.txt 2> $O/ab_7b_384.err
echo "7b384 rc=$?"; cat $O/ab_7b_384.txt | cut -c1-200
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 2048 fp8 0:0:8:0:1 0:0:8:1:0 0:0:8:0:0 0:0:16:1:0 0:0:4:1:0  > $O/ab_mistral_2k.txt 2> $O/ab_mistral_2k.err
echo "mistral2k rc=$?"; cat $O/ab_mistral_2k.txt | cut -c1-200
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 4096 fp8 0:0:0:0 0:0:8:1:0 0:0:16:1:0 > $O/ab_mistral_4k.txt 2> $O/ab_mistral_4k.err
echo "mistral4k rc=$?"; cat $O/ab_mistral_4k.txt | cut -c1-200
