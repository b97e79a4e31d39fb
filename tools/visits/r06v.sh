#!/bin/bash
# r06v: grouped matrix-core slices merging among themselves (no combine launch) — parity, A/B at Mistral 8k / 16k; new tune policy
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attention_fullgeom.py tests/test_gpu_engine.py -m gpu -x -q -k "grouped or slices or sliced or fused_launch or window or 4x2048 or mistral" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:32:0:1 0:0:32:2:1 0:0:16:1:0 0:0:16:2:1 0:0:24:2:1 0:0:32:2:1 > $O/ab_mistral_8k.txt 2> $O/ab_mistral_8k.err
echo "mistral rc=$?"; python tools/ab_print.py $O/ab_mistral_8k.txt | cat
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 4096 fp8 0:0:16:1:0 0:0:16:2:1 0:0:8:2:1 > $O/ab_mistral_4k.txt 2> $O/ab_mistral_4k.err
echo "mistral4k rc=$?"; python tools/ab_print.py $O/ab_mistral_4k.txt | cat
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 8192 fp16 0:0:32:0:1 0:0:32:2:1 0:0:16:1:0 > $O/ab_mistral_8k_fp16.txt 2> $O/ab_mistral_8k_fp16.err
echo "mistral fp16 rc=$?"; python tools/ab_print.py $O/ab_mistral_8k_fp16.txt | cat
