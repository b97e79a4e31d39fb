# PMC passes over the prompt pass (2 layers, 4 x 2048): where the prefill attention kernel's wave cycles go
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_pmc1 gpurun_out/attn_pmc2
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/attn_pmc1 -- python tools/prefill_engine_bench.py 4 2048 128 0 2 > gpurun_out/attn_pmc1.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/attn_pmc2 -- python tools/prefill_engine_bench.py 4 2048 128 0 2 > gpurun_out/attn_pmc2.log 2>&1
(python tools/pmc_summary.py gpurun_out/attn_pmc1 attn_prefill; python tools/pmc_summary.py gpurun_out/attn_pmc2 attn_prefill; python tools/pmc_summary.py gpurun_out/attn_pmc1 "gemm_f16p_kernel<0, false, 0, false"; ) > gpurun_out/attn_pmc.txt 2>&1
tail -3 gpurun_out/attn_pmc1.log gpurun_out/attn_pmc2.log | cut -c1-160
cat gpurun_out/attn_pmc.txt
