#!/bin/bash
# mkvariant.sh <name> [-D...]: libwoq_hip.so with woq_gemm_f16.hip compiled under extra switches -> tools/lib_gemm_<name>.so
set -e
cd "$(dirname "$0")/../intel_extension_for_transformers_amd/csrc"
name=$1; shift
make -j8 >/dev/null
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 -fvisibility=hidden -Wno-unused-value"
/opt/rocm/bin/hipcc $FL "$@" -c woq_gemm_f16.hip -o _build/var_$name.o
objs=$(ls _build/woq_*.o | grep -v woq_gemm_f16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _build/var_$name.o -o ../../tools/lib_gemm_$name.so
/opt/rocm/bin/hipcc $FL "$@" -S --cuda-device-only woq_gemm_f16.hip -o /tmp/var_$name.s 2>/dev/null
grep -A12 "\.name:.*gemm_f16s_kernelILi0ELb0ELb0ELi1E" /tmp/var_$name.s | grep -E "vgpr_count|vgpr_spill|\.name" | head -4
