// woq_gemv_common.h — pieces shared by the two decode-GEMV kernels (woq_gemv_i8.hip: fp32 activation rows staged in
// the kernel; woq_gemv_xq.hip: activations that arrive as pre-converted limb blocks).
#pragma once
#include "woq_device.h"
#include "woq_launch.h"

namespace woq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw (as loaded) scale words of one tile, converted where they are used
template <int SMODE, bool S32>
struct RawSc;
template <>
struct RawSc<0, false> {
  typedef uint16_t type;
};
template <>
struct RawSc<0, true> {
  typedef float type;
};
template <>
struct RawSc<1, false> {
  typedef uint2 type;
};
template <>
struct RawSc<1, true> {
  typedef float4_t type;
};

__device__ __forceinline__ float tscale16(uint32_t bits, bool is_bf16) {
  const float a = bf16_bits_to_f32((uint16_t)bits), b = f16_bits_to_f32((uint16_t)bits);
  return is_bf16 ? a : b;
}

// raw buffer descriptor (gfx950: dword3 = 0x00020000, 32-bit raw data format). Out-of-range reads return 0 and
// touch no memory, so slice / matrix edges need no clamps or masks anywhere below. `p` must be wave-uniform.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
constexpr int AUX_NT = 2;  // non-temporal: streamed-once weights

// three limb sums (D rows 4m..4m+2) + sum of 16*q (row 4m+3) of one lane -> the exact integer
// sum_k 16 q_k (Q_k - 2^22) rounded once to fp32, Q = 23-bit offset-binary activation:
//   Q - 2^22 = (b2 - 64) 2^16 + b1 2^8 + b0, rows hold b0 - 128, b1 - 128, b2
__device__ __forceinline__ float limb_combine(const i32x4& d) {
  const int i0 = d.x + (d.w << 7), i1 = d.y + (d.w << 7), i2 = d.z - (d.w << 6);
  return fmaf((float)i2, 65536.f, fmaf((float)i1, 256.f, (float)i0));
}

// geometry pick of the tile GEMVs (woq_gemv_i8.hip): nw waves x tpw tiles cover tiles_k; false = not covered
bool gemv_tile_geometry(int tiles_k, int cb, int smode, int& nw, int& tpw);
// K ranges one launch cannot hold run as chained launches; number of chunks, 0 = not covered
int gemv_tile_k_chunks(int tiles_k, int cb, int smode, bool chainable);

}  // namespace woq
