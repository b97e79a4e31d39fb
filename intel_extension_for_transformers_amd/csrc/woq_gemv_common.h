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
#ifndef WOQ_AUX_NT  // A/B builds: tools/mkvariant_xq.sh <name> -DWOQ_AUX_NT=0 (default cache policy on the weight stream)
#define WOQ_AUX_NT 2
#endif
constexpr int AUX_NT = WOQ_AUX_NT;  // non-temporal: streamed-once weights

// three limb sums (D rows 4m..4m+2) + sum of 16*q (row 4m+3) of one lane -> the exact integer
// sum_k 16 q_k (Q_k - 2^22) rounded once to fp32, Q = 23-bit offset-binary activation:
//   Q - 2^22 = (b2 - 64) 2^16 + b1 2^8 + b0, rows hold b0 - 128, b1 - 128, b2
__device__ __forceinline__ float limb_combine(const i32x4& d) {
  const int i0 = d.x + (d.w << 7), i1 = d.y + (d.w << 7), i2 = d.z - (d.w << 6);
  return fmaf((float)i2, 65536.f, fmaf((float)i1, 256.f, (float)i0));
}

// ---- 4-bit table weight types (nf4 / fp4) on the int8 MFMA (round 4) ----
// w = table[code] * scale is not linear in the code, but the table has 16 entries: table[c] * S (S = 2^22 for nf4,
// 2 for fp4_e2m1, 192 for the bitsandbytes fp4 table — exact for both fp4 tables, |error| <= 2^-23 of the table's
// largest magnitude for nf4 at S = 2^22; 127 * 256 and two digits for the reduced-precision compute modes) is an integer
// written in NDIG balanced base-256 digits, and each digit plane of a weight
// tile is a byte-table lookup of its nibbles: v_perm_b32 picks from 8 table bytes, so a 16-entry lookup of four codes is
// two perms (codes 0..7, codes 8..15) and a select by bit 3 of each code. One MFMA per digit plane against the same A
// operand, the results recombined as f0 + 2^8 f1 + 2^16 f2 — weights stay 4 bits in HBM, the inner product stays exact
// in int32, and the kernels are the int4 ones with a different B-operand unpack (12 VALU per four weights for three
// digits against 0.4 for int4: such a launch is issue-bound at ~1.5x the int4 time, against 4x for the fp32 VALU kernel
// these types ran on through round 3).
struct LutArgs {
  uint32_t d[3][4];  // digit plane j: bytes 0..15 = digit j of table[0..15] * S
  float wmul;        // 16 / S: the int4 kernels carry 16 q and fold 2^-4 into the activation factor
};

// digits of the four codes held one per byte in idx (values 0..15)
template <int NDIG>
__device__ __forceinline__ void lut_quad(const LutArgs& L, uint32_t idx, uint32_t (&out)[NDIG]) {
  const uint32_t sel = idx & 0x07070707u;
#ifndef WOQ_LUT_ARITH_MASK
  // 0xff in every byte whose code is >= 8, from v_perm_b32's constant selectors: 0x05 picks byte 1 of the (zero) high
  // source, 0x0d yields 0xff (ISA: selector 12 = 0x00, above = 0xff). Two instructions; the arithmetic form below takes
  // three: +1.7 % / +2.1 % tokens/s for three / two planes at the 7B shape (profiles/r04ae_last_tree_summary.txt)
  const uint32_t mask = __builtin_amdgcn_perm(0u, 0u, (idx & 0x08080808u) | 0x05050505u);
#else
  const uint32_t m8 = idx & 0x08080808u;
  const uint32_t mask = (m8 << 5) - (m8 >> 3);
#endif
#pragma unroll
  for (int j = 0; j < NDIG; ++j) {
    const uint32_t lo = __builtin_amdgcn_perm(L.d[j][1], L.d[j][0], sel);  // {hi:lo} = table bytes 7..0
    const uint32_t hi = __builtin_amdgcn_perm(L.d[j][3], L.d[j][2], sel);  // table bytes 15..8
    out[j] = (hi & mask) | (lo & ~mask);
  }
}
// B operands (one per digit plane) of the 64-k half whose nibbles sit in w0, w1 — same k order as the int4 unpack:
// low nibbles of w0, high nibbles of w0, low nibbles of w1, high nibbles of w1
template <int NDIG>
__device__ __forceinline__ void lut_b(const LutArgs& L, uint32_t w0, uint32_t w1, i32x4 (&b)[NDIG]) {
  uint32_t q0[NDIG], q1[NDIG], q2[NDIG], q3[NDIG];
  lut_quad<NDIG>(L, w0 & 0x0f0f0f0fu, q0);
  lut_quad<NDIG>(L, (w0 >> 4) & 0x0f0f0f0fu, q1);
  lut_quad<NDIG>(L, w1 & 0x0f0f0f0fu, q2);
  lut_quad<NDIG>(L, (w1 >> 4) & 0x0f0f0f0fu, q3);
#pragma unroll
  for (int j = 0; j < NDIG; ++j) b[j] = i32x4{(int)q0[j], (int)q1[j], (int)q2[j], (int)q3[j]};
}
__device__ __forceinline__ void int4_b(uint32_t w0, uint32_t w1, i32x4& b) {
  b = i32x4{(int)((w0 << 4) & 0xf0f0f0f0u), (int)(w0 & 0xf0f0f0f0u), (int)((w1 << 4) & 0xf0f0f0f0u),
            (int)(w1 & 0xf0f0f0f0u)};
}
// host: the digit planes of a table weight type; returns NDIG (1 | 2 | 3), 0 = not a table. compute_type (woq_blob.h):
// nf4 takes three planes for fp32 compute and two (table held to 2^-16 of its largest entry — finer than the bf16 / fp16
// operands those modes ask for) otherwise
int lut_args_for(uint32_t weight_type, uint32_t compute_type, LutArgs& L);

// geometry pick of the tile GEMVs (woq_gemv_i8.hip): nw waves x tpw tiles cover tiles_k; false = not covered
bool gemv_tile_geometry(int tiles_k, int cb, int smode, int& nw, int& tpw);
// K ranges one launch cannot hold run as chained launches; number of chunks, 0 = not covered
int gemv_tile_k_chunks(int tiles_k, int cb, int smode, bool chainable);

}  // namespace woq
