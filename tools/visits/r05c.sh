#!/bin/bash
# r05c: overlap probe variants (who polls, how often) with timelines; act-order engine tests after the row-order fix
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
for cfg in "0 0 8" "0 1 8" "0 1 2" "0 1 32" "1500 1 8" "0 0 32"; do
  timeout 60 tools/overlap_probe.bin 32 5 $cfg >> $O/overlap.txt 2>&1; echo "overlap $cfg rc=$?"
done
cat $O/overlap.txt
timeout 600 python -m pytest tests/test_gpu_api.py -q -m gpu -k "gptq or desc_act" > $O/pytest_gptq.txt 2>&1; tail -5 $O/pytest_gptq.txt
