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

// bit shift of k-offset j (0..15 of a lane's 16-k run in one 64-k half) inside its packed u32 — the device twin of
// woq_nibble_shift() in include/woq_blob.h: byte j & 3, low nibble for (j & 7) < 4, high nibble otherwise.
__host__ __device__ __forceinline__ int nibble_shift(int j) { return 8 * (j & 3) + 4 * ((j & 7) >> 2); }

// wave64 sum over the 4 lanes that share a column (lane, lane^16, lane^32, lane^48)
__device__ __forceinline__ float reduce_kq(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// wave64 all-reduce without LDS traffic: four DPP butterfly steps leave every 16-lane row uniform
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row values are combined through SGPRs.
// ~10 short instructions instead of six dependent ds_bpermute round trips (~100 cycles each).
#define WOQ_DPP_F32(v, ctrl) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += WOQ_DPP_F32(v, 0xB1);
  v += WOQ_DPP_F32(v, 0x4E);
  v += WOQ_DPP_F32(v, 0x141);
  v += WOQ_DPP_F32(v, 0x140);
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) +
          __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) +
          __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, WOQ_DPP_F32(v, 0xB1));
  v = fmaxf(v, WOQ_DPP_F32(v, 0x4E));
  v = fmaxf(v, WOQ_DPP_F32(v, 0x141));
  v = fmaxf(v, WOQ_DPP_F32(v, 0x140));
  const int b = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
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


// ---- OCP e4m3fn (KV-cache storage). Conversions saturate at +-448 (the format has no infinity). ----
__device__ __forceinline__ uint32_t f32x2_to_fp8x2(float a, float b) {
  a = fminf(fmaxf(a, -448.f), 448.f);
  b = fminf(fmaxf(b, -448.f), 448.f);
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
struct Fp8 {  // one cache element
  uint8_t b;
  __device__ Fp8() = default;
  __device__ explicit Fp8(float f) : b((uint8_t)(f32x2_to_fp8x2(f, 0.f) & 0xffu)) {}
  __device__ explicit operator float() const { return __builtin_amdgcn_cvt_f32_fp8((int)b, 0); }
};
struct __attribute__((aligned(8))) Fp8x8 {  // eight consecutive cache elements
  uint32_t lo, hi;
  __device__ Fp8 operator[](int i) const {
    Fp8 r;
    r.b = (uint8_t)(((i < 4 ? lo : hi) >> (8 * (i & 3))) & 0xffu);
    return r;
  }
};
template <typename KV>
struct KvVec8 {
  typedef KV type __attribute__((ext_vector_type(8)));
};
template <>
struct KvVec8<Fp8> {
  typedef Fp8x8 type;
};
// eight cache elements -> fp32. e4m3: four v_cvt_pk_f32_fp8 (two values per instruction, the word half is an operand
// modifier) instead of a byte extraction + v_cvt_f32_fp8 per element — same values, a third of the VALU work
// (round 6: the per-head decode attention over an fp8 cache is VALU-bound at long contexts)
template <typename V8>
__device__ __forceinline__ void kv8_to_f32(const V8& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ void kv8_to_f32(const Fp8x8& v, float (&f)[8]) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.lo, true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.hi, true);
  f[0] = a.x, f[1] = a.y, f[2] = b.x, f[3] = b.y, f[4] = c.x, f[5] = c.y, f[6] = d.x, f[7] = d.y;
}


// ---- 4-bit table weight types (woq_blob.h): w = table[code] * scale ----
__host__ __device__ __forceinline__ bool is_table_type(uint32_t t) { return t >= 2u && t <= 4u; }
__host__ __device__ __forceinline__ float lut_max(uint32_t t) { return t == 3u ? 6.0f : 1.0f; }
__device__ __forceinline__ float lut_value(uint32_t weight_type, int code) {
  constexpr float nf4[16] = WOQ_LUT_NF4;
  constexpr float e2m1[16] = WOQ_LUT_FP4_E2M1;
  constexpr float bnb[16] = WOQ_LUT_FP4_BNB;
  return weight_type == WOQ_W_NF4 ? nf4[code] : (weight_type == WOQ_W_FP4_E2M1 ? e2m1[code] : bnb[code]);
}

// ---- fp8 weight types (woq_blob.h): w = value(code) * scale ----
__host__ __device__ __forceinline__ bool is_fp8_type(uint32_t t) { return t == 7u || t == 8u; }
// OCP e4m3fn (no infinities, S.1111.111 = NaN) / e5m2 (IEEE-like: exponent 31 = inf / NaN), by the definition
__host__ __device__ __forceinline__ float fp8_code_value(uint32_t weight_type, int code) {
  const int s = (code >> 7) & 1;
  float v;
  if (weight_type == WOQ_W_FP8_E4M3) {
    const int e = (code >> 3) & 15, m = code & 7;
    if (e == 15 && m == 7)
      v = NAN;
    else
      v = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10);
  } else {
    const int e = (code >> 2) & 31, m = code & 3;
    if (e == 31)
      v = m == 0 ? INFINITY : NAN;
    else
      v = e == 0 ? ldexpf((float)m, -16) : ldexpf((float)(4 + m), e - 17);
  }
  return s ? -v : v;
}

}  // namespace woq
