#!/bin/bash
# mkvariant_xq.sh <name> [-D...]: libwoq_hip.so with the XQ decode GEMV sources (woq_gemv_xq.hip, woq_gemv_attn.hip)
# compiled under extra switches -> tools/lib_xq_<name>.so; select with WOQ_HIP_LIB=<path> (same-box A/B runs)
set -e
cd "$(dirname "$0")/../intel_extension_for_transformers_amd/csrc"
name=$1; shift
make -j8 >/dev/null
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 -fvisibility=hidden -Wno-unused-value"
for f in woq_gemv_xq woq_gemv_attn; do /opt/rocm/bin/hipcc $FL "$@" -c $f.hip -o _build/varxq_${name}_$f.o; done
objs=$(ls _build/woq_*.o | grep -v "woq_gemv_xq.o\|woq_gemv_attn.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _build/varxq_${name}_woq_gemv_xq.o _build/varxq_${name}_woq_gemv_attn.o -o ../../tools/lib_xq_$name.so
echo built tools/lib_xq_$name.so
