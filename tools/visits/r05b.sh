#!/bin/bash
# r05b: (1) can dependent launches overlap heads / tails? any-order flag and rotating streams vs serial launches on a
# chain of load-only kernels shaped like the 7B layer (tools/overlap_probe.hip); (2) act-order (g_idx) decode on the
# tile GEMV's gather form: parity tests + the HF GPTQ checkpoint tests; (3) the bench-line plumbing touched this round
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
for work in 0 1500; do
  timeout 60 tools/overlap_probe.bin 32 5 $work >> $O/overlap.txt 2>&1; echo "overlap work=$work rc=$?"
done
cat $O/overlap.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "act_order or decode_vs_oracle or packed_weight_info" -x > $O/pytest_shuf.txt 2>&1; tail -3 $O/pytest_shuf.txt
timeout 600 python -m pytest tests/test_gpu_api.py -q -m gpu -k "gptq or desc_act" -x > $O/pytest_gptq.txt 2>&1; tail -3 $O/pytest_gptq.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prefill-seqs 0 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?"; tail -c 1500 $O/bench20.json
