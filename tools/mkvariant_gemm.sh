#!/bin/bash
# mkvariant_gemm.sh <name> [-D...]: libwoq_hip.so with woq_gemm_f16.hip compiled under extra switches -> tools/lib_gemm_<name>.so
set -e
cd "$(dirname "$0")/../intel_extension_for_transformers_amd/csrc"
name=$1; shift
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 -fvisibility=hidden -Wno-unused-value"
/opt/rocm/bin/hipcc $FL "$@" -c woq_gemm_f16.hip -o _build/vargemm_${name}.o
objs=$(ls _build/woq_*.o | grep -v "woq_gemm_f16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _build/vargemm_${name}.o -o ../../tools/lib_gemm_$name.so
echo built tools/lib_gemm_$name.so
