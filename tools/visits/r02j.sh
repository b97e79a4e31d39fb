#!/bin/bash
# the N > 1 code path of bench.py (Llama-2-70B at TP = N) with N ranks on ONE GPU over gloo: validates the launch
# contract, the device exchange inside the captured graph, token agreement — not a performance number
set -u
OUT=gpurun_out/r02j; mkdir -p $OUT
for N in 2 4; do
WOQ_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $N --steps 16 --warmup 4 --layers 16 > $OUT/tp$N.json 2> $OUT/tp$N.err; echo "tp$N rc=$?"; tail -c 1500 $OUT/tp$N.json; tail -5 $OUT/tp$N.err
done
WOQ_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 8 --warmup 2 --layers 8 --host-allreduce > $OUT/tp2_host.json 2> $OUT/tp2_host.err; echo "tp2 host rc=$?"; tail -c 600 $OUT/tp2_host.json; tail -5 $OUT/tp2_host.err
