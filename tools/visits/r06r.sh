#!/bin/bash
# r06r: + the combine as the first workgroups of the o_proj launch — parity, then same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_attention_fullgeom.py tests/test_gpu_engine.py -m gpu -x -q -k "fused_launch or sliced or slices or 4x2048 or window or mistral" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
# variants fold:chunk:splits:fs:grouped
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 2048 fp16 0:0:16:0 0:0:8:1 0:0:8:3 0:0:6:3 0:0:4:3 0:0:16:3 0:0:16:2 > $O/ab_7b_2048.txt 2> $O/ab_7b_2048.err
echo "7b rc=$?"; cat $O/ab_7b_2048.txt | cut -c1-200
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:0:0 0:0:0:2 0:0:16:3:0 0:0:8:3:0 0:0:16:2 0:0:24:2 0:0:0:0 > $O/ab_mistral_8k.txt 2> $O/ab_mistral_8k.err
echo "mistral rc=$?"; cat $O/ab_mistral_8k.txt | cut -c1-200
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 512 fp16 0:0:8:0 0:0:4:1 0:0:4:3 0:0:2:3 0:0:3:3 0:0:1:1 > $O/ab_7b_512.txt 2> $O/ab_7b_512.err
echo "7b512 rc=$?"; cat $O/ab_7b_512.txt | cut -c1-200
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 256 fp16 0:0:1:1 0:0:2:3 0:0:4:3 > $O/ab_7b_256.txt 2> $O/ab_7b_256.err
echo "7b256 rc=$?"; cat $O/ab_7b_256.txt | cut -c1-200
