#!/bin/bash
set -u
OUT=gpurun_out/r02q; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python tools/tp_shard_bench.py 80 > $OUT/tp_shard.json 2>$OUT/tp_shard.err; cat $OUT/tp_shard.json
