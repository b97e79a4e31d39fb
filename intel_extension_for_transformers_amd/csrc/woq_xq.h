// woq_xq.h — "XQ": an activation vector already in the decode GEMV's A-operand form.
//
// Why. gemv_tile_kernel (woq_gemv_i8.hip) converts its fp32 activation row to three int8 limb rows inside EVERY
// workgroup (768 / 256 / 688 / 256 times per layer for the Llama-2-7B projections) — loads of x and the norm weight,
// a wave-wide max and sum, ~170 VALU and 12 LDS stores per wave, all on the launch's critical path. Knocking that
// stage out of the kernel saves 1.8 / 0.9 / 1.6 / 1.5 us of the 8.5 / 5.2 / 12.1 / 8.2 us back-to-back launch times
// (profiles/r02b_gemv_knockouts.txt): 17 % of the GEMV time is re-deriving the same bytes. XQ moves the conversion to
// the PRODUCER of the vector — 16 threads of the epilogue that already hold the 16 values — so it happens once.
//
// Format, for K = 16 * nb values: block b = values [16 b, 16 b + 16) — exactly what one producer workgroup (one
// 16-column output tile of the previous GEMV, or 16 lanes of the attention / embedding kernels) owns, so no
// cross-workgroup reduction is needed:
//   limbs [nb][3][16] int8 : BALANCED signed digits of v = round(y * 2^(21 - e_b)) (|v| <= 2^22), e_b = exponent of
//                            the block's max |y|: v = s0 + 2^8 s1 + 2^16 s2 with s0, s1 in [-128, 127], s2 in
//                            [-64, 64] (round 3; round 2 stored offset-binary bytes and needed a ones row in the MFMA
//                            to take the offsets out again — 9 VALU per MFMA result instead of 4)
//   u     [nb] fp32        : 2^(e_b - 25), undoes the block's fixed-point scale and the 16 * q weight bytes
//   sx    [nb] fp32        : sum over the block of v, the fixed-point activations' sum (zero-point term)
// A block is one (64-k half, lane quarter) of the consumer's v_mfma_i32_16x16x64_i8: MFMA rows 4 e .. 4 e + 2 carry
// quarter e's three digits (row 4 e + 3 and the other quarters' k are zero), so ONE MFMA per half returns the four
// blocks' digit sums separately, one per lane quarter, each scaled by its own u (and, for group 32, its own weight
// scale). Element error <= max|x_block| * 2^-22. An all-zero buffer is a valid vector of zeros.
// RMSNorm stays separable: the producer multiplies by the NEXT norm's weight before converting and leaves one partial
// sum of squares of the raw values per block; the consumer adds them up in a fixed order.
#pragma once
#include "woq_device.h"

namespace woq {

struct XqPtrs {  // device pointers of one XQ vector (all null = absent)
  uint8_t* limbs;
  float* u;
  float* sx;
};

// bytes of an XQ vector of K values: limbs padded so that 1-KiB LDS-DMA pieces never leave the allocation
__host__ __device__ inline size_t xq_limb_bytes(int K) { return (((size_t)(K / 16) * 48 + 1023) / 1024) * 1024 + 1024; }
__host__ __device__ inline size_t xq_bytes(int K) { return xq_limb_bytes(K) + (size_t)(K / 16) * 8; }
__host__ inline XqPtrs xq_carve(void* base, int K) {
  XqPtrs p;
  p.limbs = (uint8_t*)base;
  p.u = (float*)((uint8_t*)base + xq_limb_bytes(K));
  p.sx = p.u + K / 16;
  return p;
}

// 16-lane (one DPP row) all-reduces: quad_perm xor 1, xor 2, row_half_mirror, row_mirror
#define WOQ_DPP_I32(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, WOQ_DPP_F32(v, 0xB1));
  v = fmaxf(v, WOQ_DPP_F32(v, 0x4E));
  v = fmaxf(v, WOQ_DPP_F32(v, 0x141));
  return fmaxf(v, WOQ_DPP_F32(v, 0x140));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += WOQ_DPP_F32(v, 0xB1);
  v += WOQ_DPP_F32(v, 0x4E);
  v += WOQ_DPP_F32(v, 0x141);
  return v + WOQ_DPP_F32(v, 0x140);
}
__device__ __forceinline__ int row16_sum_i32(int v) {
  v += WOQ_DPP_I32(v, 0xB1);
  v += WOQ_DPP_I32(v, 0x4E);
  v += WOQ_DPP_I32(v, 0x141);
  return v + WOQ_DPP_I32(v, 0x140);
}

// In-launch hand-off of an XQ vector (round 3, the chained launches — tools/rejected/woq_gemv_chain.hip — and the persistent launch): a consumer workgroup of the
// SAME launch may be waiting for block `blk`. The producer then stores the block write-through (agent-scope stores
// never stay in this XCD's L2 alone), drains them, and only then stores the block's flag word = the (step, vector) tag;
// the consumer polls the flags of its K slice and reads the blocks with agent-scope loads afterwards
// (cdna_hip_programming.md Guideline 16, form R1). flag == nullptr: plain stores, the consumer is a later launch.
struct XqPub {
  unsigned int* flag;  // [K / 16] one word per block, or null
  unsigned int tag;    // never 0 (flags rest at 0 / at an older tag)
};

// Called by the 16 lanes of ONE DPP row (lanes 16 r .. 16 r + 15, all active), lane j = lane & 15 holding value
// 16 * blk + j of the vector: writes block `blk`. Conversion as gemv_tile_kernel's stage_row (fp32 sum with
// 1.5 * 2^23: the mantissa IS round(y * 2^(21 - e)) + 2^22), then three balanced digits.
// ssq_out / ss (optional): the block's sum of squares of the raw values, published with the block.
template <bool PUBLISH = false>
__device__ __forceinline__ void xq_emit16(float y, const XqPtrs& o, int blk, int j, const XqPub& pub = XqPub{nullptr, 0u},
                                          float* ssq_out = nullptr, float ss = 0.f) {
  const float amax = row16_max(fabsf(y));
  int e = 0;
  if (amax > 0.f && amax < INFINITY) e = max(-100, min(100, __builtin_amdgcn_frexp_expf(amax)));
  const uint32_t Q = __float_as_uint(fmaf(y, ldexpf(1.f, 21 - e), 12582912.f));
  const int v = (int)(Q & 0x7fffffu) - (1 << 22);
  const int s0 = (int)((uint32_t)v << 24) >> 24;  // low byte, sign-extended
  const int v1 = (v - s0) >> 8;
  const int s1 = (int)((uint32_t)v1 << 24) >> 24;
  const int s2 = (v1 - s1) >> 8;
  uint8_t* d = o.limbs + (size_t)blk * 48 + j;
  const int qs = row16_sum_i32(v);
  if constexpr (PUBLISH) {
    __hip_atomic_store(d, (uint8_t)s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 16, (uint8_t)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 32, (uint8_t)s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j == 0) {
      __hip_atomic_store((unsigned int*)(o.u + blk), __float_as_uint(ldexpf(1.f, e - 25)), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned int*)(o.sx + blk), __float_as_uint((float)qs), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      if (ssq_out != nullptr)
        __hip_atomic_store((unsigned int*)(ssq_out + blk), __float_as_uint(ss), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the block has landed (every storing lane is in this wave)
    if (j == 0) __hip_atomic_store(pub.flag + blk, pub.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    d[0] = (uint8_t)s0;
    d[16] = (uint8_t)s1;
    d[32] = (uint8_t)s2;
    if (j == 0) {
      o.u[blk] = ldexpf(1.f, e - 25);
      o.sx[blk] = (float)qs;
      if (ssq_out != nullptr) ssq_out[blk] = ss;
    }
  }
}

}  // namespace woq
