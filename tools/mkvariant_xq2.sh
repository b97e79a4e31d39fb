#!/bin/bash
# mkvariant_xq2.sh <name> "<flags for woq_gemv_xq.hip>" "<flags for woq_gemv_attn.hip>": libwoq_hip.so with the two XQ
# decode GEMV sources compiled under (different) extra switches -> tools/lib_xq_<name>.so; select with WOQ_HIP_LIB=<path>
set -e
cd "$(dirname "$0")/../intel_extension_for_transformers_amd/csrc"
name=$1
make -j8 >/dev/null
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 -fvisibility=hidden -Wno-unused-value"
/opt/rocm/bin/hipcc $FL $2 -c woq_gemv_xq.hip -o _build/varxq_${name}_woq_gemv_xq.o
/opt/rocm/bin/hipcc $FL $3 -c woq_gemv_attn.hip -o _build/varxq_${name}_woq_gemv_attn.o
objs=$(ls _build/woq_*.o | grep -v "woq_gemv_xq.o\|woq_gemv_attn.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _build/varxq_${name}_woq_gemv_xq.o _build/varxq_${name}_woq_gemv_attn.o -o ../../tools/lib_xq_$name.so
echo built tools/lib_xq_$name.so
