#!/bin/bash
# r06m: mid-round records of the tree — full GPU suite, smoke, the driver's bench command, kernel trace of the bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"; cat $O/bench20.json; cp bench_extra.json $O/bench20_extra.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity > $O/bench_prof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
python tools/prof_stats.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) 16 > $O/kernel_stats.txt 2>&1; cat $O/kernel_stats.txt | cut -c1-170; rm -rf $O/prof
