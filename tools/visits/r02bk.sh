#!/bin/bash
# r02bk: LDS bank conflicts of the ring GEMM (half-tile layout, ht_swz)
mkdir -p gpurun_out/r02bk; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/r02bk/p -o p --output-format csv -- $R/tools/gemm_probe.bin $R/intel_extension_for_transformers_amd/libwoq_hip.so 8192 4096 22016 128 0 bf16 2 > $R/gpurun_out/r02bk/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r02bk/p/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm_f16' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k, 'n=%d mean=%.4g' % (len(v), sum(v) / len(v)))
PY
