#!/bin/bash
# r02at: prompt-pass attention with the mask work under a wave-uniform branch (new) against the previous build (old)
L=intel_extension_for_transformers_amd/libwoq_hip.so
cp $L /tmp/new.so
for lib in new old new old; do
  [ $lib = old ] && cp tools/lib_gemm_cur.so $L || cp /tmp/new.so $L
  echo "$lib 4x2048: $(timeout 300 python tools/prefill_engine_bench.py 4 2048 128 0 2>/dev/null | tail -1 | sed 's/.*"s": //; s/, "linear.*mfma_frac_of_2500": / frac /; s/}//')"
  echo "$lib mistral 8k: $(timeout 300 python tools/prefill_engine_bench.py 1 8192 128 0 32 8 14336 2048 2>/dev/null | tail -1 | sed 's/.*"s": //; s/, "linear.*mfma_frac_of_2500": / frac /; s/}//')"
done
cp /tmp/new.so $L
