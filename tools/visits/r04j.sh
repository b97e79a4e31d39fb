#!/bin/bash
# r04j: combine in one round trip + prologue requests ahead of the K/V burst: tests, long-context timing + kernel trace, PMC traffic pass, TP rank kernel trace
set -u
TAG=r04j; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_attention_fullgeom.py -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python tools/longctx_ab.py 16 8192 fp8 0:0:0 0:0:0 > $OUT/longctx.txt 2>&1; grep fold $OUT/longctx.txt | cut -c1-200
LCAB_KVH=32 LCAB_INTER=11008 timeout 600 python tools/longctx_ab.py 16 2048 fp16 0:0:0 > $OUT/longctx_mha2k.txt 2>&1; grep fold $OUT/longctx_mha2k.txt | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_lc -o lc -- python tools/longctx_ab.py 8 8192 fp8 0:0:0 > $OUT/prof_lc.txt 2>&1
DB=$(ls $OUT/prof_lc/*.db $OUT/prof_lc/*/*.db 2>/dev/null | head -1)
python tools/prof_stats.py $DB 9 2>&1 | cut -c1-60,92-170 | tee $OUT/kernels_longctx.txt; rm -rf $OUT/prof_lc
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_tp -o tp -- python tools/tp_shard_bench.py 16 > $OUT/prof_tp.txt 2>&1
DB=$(ls $OUT/prof_tp/*.db $OUT/prof_tp/*/*.db 2>/dev/null | head -1)
python tools/prof_stats.py $DB 9 2>&1 | cut -c1-60,92-170 | tee $OUT/kernels_tp_rank.txt; rm -rf $OUT/prof_tp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-parity --prefill-seqs 0 --no-structures --condition-ms 0 > $OUT/pmc_bench.json 2> $OUT/pmc.err; echo "pmc rc=$?"
python tools/pmc_traffic.py $OUT/pmc $OUT/pmc_traffic.json gemv_xqs > $OUT/pmc_summary.txt 2>&1; tail -12 $OUT/pmc_summary.txt | cut -c1-200
rm -rf $OUT/pmc
