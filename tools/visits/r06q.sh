#!/bin/bash
# r06q: context slices inside the fused qkv launch — parity tests, then same-box A/B at 7B @ 2048 (fp16) and Mistral @ 8k (fp8)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention_fullgeom.py -m gpu -x -q -k "fused_launch or sliced_regime or 4x2048" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
# variants fold:chunk:splits:fs:grouped
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 2048 fp16 0:0:16:0 0:0:16:1 0:0:8:1 0:0:32:1 0:0:8:0 0:0:16:1 > $O/ab_7b_2048.txt 2> $O/ab_7b_2048.err
echo "7b rc=$?"; cat $O/ab_7b_2048.txt | cut -c1-230
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:0:0 0:0:32:1:0 0:0:16:1:0 0:0:64:1:0 0:0:32:0:0 0:0:0:0 > $O/ab_mistral_8k.txt 2> $O/ab_mistral_8k.err
echo "mistral rc=$?"; cat $O/ab_mistral_8k.txt | cut -c1-230
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 512 fp16 0:0:8:0 0:0:8:1 0:0:4:1 0:0:1:1 > $O/ab_7b_512.txt 2> $O/ab_7b_512.err
echo "7b512 rc=$?"; cat $O/ab_7b_512.txt | cut -c1-230
