#!/bin/bash
# r03s: the N > 1 bench path with two ranks as two processes on ONE GPU (gloo rendezvous, device exchange over hipIpc to
# self): tensor-parallel XQ kernels + fused push against the fp32-activation kernels / kernel push. 16 layers of the 70B shape.
OUT=gpurun_out/r03s; mkdir -p $OUT
i=0
for mode in "1 1" "1 0" "0 0"; do
  set -- $mode
  i=$((i+1))
  WOQ_TP_XQ=$1 WOQ_TP_FUSED_PUSH=$2 WOQ_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29530+i)) bench.py --gpus 2 --steps 32 --warmup 4 --layers 16 > $OUT/run$i.log 2>&1
  echo "WOQ_TP_XQ=$1 WOQ_TP_FUSED_PUSH=$2: $(grep '^{' $OUT/run$i.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tokens/s', d['ms_per_step'], 'ms', d['config']['allreduce_transport'][:50], d['config'].get('rccl_ranks_verified'))")"
done
