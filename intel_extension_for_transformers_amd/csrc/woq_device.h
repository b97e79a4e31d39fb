// woq_device.h — device-side helpers shared by the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/woq_blob.h"

namespace woq {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even, as bestla_customop.hpp's bf16 store
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

__device__ __forceinline__ float load_f32(const void* p, size_t i, int dt) {
  if (dt == WOQ_F32) return ((const float*)p)[i];
  uint16_t b = ((const uint16_t*)p)[i];
  return dt == WOQ_BF16 ? bf16_bits_to_f32(b) : f16_bits_to_f32(b);
}
__device__ __forceinline__ void store_f32(void* p, size_t i, int dt, float v) {
  if (dt == WOQ_F32)
    ((float*)p)[i] = v;
  else if (dt == WOQ_BF16)
    ((uint16_t*)p)[i] = f32_to_bf16_bits(v);
  else
    ((uint16_t*)p)[i] = f32_to_f16_bits(v);
}

// position of k-offset j (0..7) inside a packed u32: nibble index pos(j) = (j >> 1) | ((j & 1) << 2).
// With this interleave `w & 0x000f000f` yields (j0 | j1 << 16), `(w >> 4) & ..` (j2, j3), `(w >> 8)`
// (j4, j5), `(w >> 12)` (j6, j7): consecutive-k pairs land in one register in MFMA fragment order.
__host__ __device__ __forceinline__ int nibble_pos(int j) { return (j >> 1) | ((j & 1) << 2); }

// wave64 sum over the 4 lanes that share a column (lane, lane^16, lane^32, lane^48)
__device__ __forceinline__ float reduce_kq(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace woq
