#!/bin/bash
# r06u: where do self-merging slices inside the fused launch start to pay? 7B shape, 16 layers, short contexts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
export TMPDIR=/tmp
for ctx in 40 96 160 224; do
LCAB_GRAPH=1 LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 $ctx fp16 0:0:1:1 0:0:2:1 0:0:4:1 0:0:3:1 0:0:1:1 > $O/ab_7b_$ctx.txt 2> $O/ab_7b_$ctx.err
echo "7b $ctx rc=$?"
done
LCAB_GRAPH=1 timeout 600 python tools/longctx_ab.py 16 128 fp16 0:0:1:1 0:0:2:1 0:0:4:1 > $O/ab_mistral_128.txt 2> $O/ab_mistral_128.err
