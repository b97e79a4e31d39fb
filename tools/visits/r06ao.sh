#!/bin/bash
# r06ao: records of the tree — full GPU suite, smoke, the bench command, kernel trace + FETCH_SIZE pass of the bench, and
# kernel trace + FETCH_SIZE pass of the long-context decode (7B shape at 2048 positions, Mistral shape at 8k fp8 KV)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ao; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"; cat $O/bench20.json; cp bench_extra.json $O/bench20_extra.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extra --no-parity > $O/bench_prof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
python tools/prof_stats.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) 16 > $O/kernel_stats.txt 2>&1; cat $O/kernel_stats.txt | cut -c1-170; rm -rf $O/prof
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python bench.py --gpus 1 --steps 20 --warmup 5 --counters-run > $O/pmc_bench.json 2> $O/pmc.err; echo "pmc rc=$?"
python tools/pmc_traffic.py $O/pmc $O/pmc_traffic.json gemv_xqs > $O/pmc_summary.txt 2>&1; tail -14 $O/pmc_summary.txt | cut -c1-220; rm -rf $O/pmc
# long context: 7B shape, 2048 cached positions, fp16 cache (args: ctx chunk kv splits group asym inter kv_heads)
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof7 -o lc -- python tools/longctx_bench.py 2048 2048 fp16 0 128 0 11008 32 > $O/lc_7b_2048.json 2> $O/lc7.err; echo "lc7 rc=$?"; cat $O/lc_7b_2048.json
python tools/prof_stats.py $(ls $O/prof7/*.db $O/prof7/*/*.db 2>/dev/null | head -1) 10 > $O/lc_7b_2048_kernel_stats.txt 2>&1; cut -c1-170 $O/lc_7b_2048_kernel_stats.txt; rm -rf $O/prof7
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc7 -- python tools/longctx_bench.py 2048 2048 fp16 0 128 0 11008 32 > /dev/null 2> $O/pmc7.err; echo "pmc7 rc=$?"
python tools/pmc_summary.py $O/pmc7 gemv_xqs_attn > $O/lc_7b_2048_pmc.txt 2>&1; cat $O/lc_7b_2048_pmc.txt | cut -c1-200; rm -rf $O/pmc7
# Mistral shape, 8192 cached positions, fp8 cache
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof8 -o lc -- python tools/longctx_bench.py 8192 2048 fp8 > $O/lc_mistral_8k.json 2> $O/lc8.err; echo "lc8 rc=$?"; cat $O/lc_mistral_8k.json
python tools/prof_stats.py $(ls $O/prof8/*.db $O/prof8/*/*.db 2>/dev/null | head -1) 10 > $O/lc_mistral_8k_kernel_stats.txt 2>&1; cut -c1-170 $O/lc_mistral_8k_kernel_stats.txt; rm -rf $O/prof8
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc8 -- python tools/longctx_bench.py 8192 2048 fp8 > /dev/null 2> $O/pmc8.err; echo "pmc8 rc=$?"
python tools/pmc_summary.py $O/pmc8 gemv_xqs_attn > $O/lc_mistral_8k_pmc.txt 2>&1; cat $O/lc_mistral_8k_pmc.txt | cut -c1-200; rm -rf $O/pmc8
