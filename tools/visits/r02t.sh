#!/bin/bash
# r02t: SQ counters of the prefill GEMM (probe binary, gate/up shape)
mkdir -p gpurun_out/r02t; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/r02t/counters.txt 2>&1
LIB=${1:-ct2}
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/r02t/$n -o p --output-format csv -- $R/tools/gemm_probe.bin $R/tools/lib_gemm_$LIB.so 8192 4096 22016 128 0 bf16 2 > $R/gpurun_out/r02t/$n.log 2>&1
}
run p1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run p2 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU
run p3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
run p4 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
cd $R/gpurun_out/r02t; ls -R . | head -40
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('p*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm_f16' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f.split('/')[0], k, 'n=%d mean=%.4g' % (len(v), sum(v) / len(v)))
PY
