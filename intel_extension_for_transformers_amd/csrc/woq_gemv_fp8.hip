// woq_gemv_fp8.hip — decode rows (1..8) of the 8-bit float weight types (fp8_e4m3 / fp8_e5m2) on the fp8 matrix cores.
//
// Replaces, for these weight types, the arithmetic behind qbits.woq_linear at small M (qbits.cpp:113-140 ->
// bestla_weightonly_dispatcher.cpp:120-189; weight strings "fp8_e4m3" / "fp8_e5m2", dispatcher.hpp:62-72). Parity
// definition: autograd/functions.py:41-63 (dequantise -> fp32 matmul -> + bias).
//
// Why a kernel of its own. w = value(code) * scale with a 256-entry value table: the int8-MFMA route of the integer and
// 4-bit table types (woq_gemv_common.h) does not apply — v_perm_b32 looks up 16 entries, not 256, and the fp32 VALU kernel
// these types ran on (woq_gemv.hip: one LDS lookup + one FMA per weight and row) takes 35-90 us per Llama-2-7B projection.
// But gfx950's matrix cores read OCP fp8 DIRECTLY: v_mfma_f32_16x16x32_fp8_fp8 (and _fp8_bf8 for e5m2 codes) takes the
// code bytes as its B operand, no table at all. What has to change is the ACTIVATION side: the A operand is fp8 too, and
// an fp32 activation is not. It goes in as SIX balanced base-16 digits: v = round(x * 2^(21 - e)) (|v| <= 2^22, e = the
// exponent of the wave slice's largest |x|, as in woq_gemv_i8.hip) = sum_j d_j 16^j with d_j in [-8, 7] — every digit is
// an exact e4m3 value — six MFMA rows per activation row, two activation rows per 16-row MFMA. Every product
// digit x weight is exact (4 x 4 significant bits), the sums run in the MFMA's fp32 accumulator, and the six digit sums
// are recombined as sum_j 16^j S_j in fp32 (no cancellation: the digits are balanced, so the top digit carries the
// magnitude), the activation held to 2^-22 of its slice maximum. Measured accuracy (profiles/r04ah_fp8_decode.txt): inside
// 2e-6 of sum |x||w| on RTN-quantised weights, 4.6e-6 on a matrix holding every finite e4m3 code uniformly — the matrix
// core aligns the 32 products of a dot to the largest one and keeps ~16 bits below it (a numpy model of this kernel with
// that window reproduces both figures: tools/fp8_mfma_accumulation_model.py); three orders of magnitude under the 2^-4
// grid of the weights themselves.
// B operand: code = (hi nibble << 4) | (lo nibble ^ 8) from the blob's two nibble planes (include/woq_blob.h
// woq_fp8_headers), both in the int4 tile layout, so four codes are assembled by four VALU from one dword of each plane:
// ~1 VALU per weight against ~5 for the lookup kernel.
// Schedule: as woq_gemv_i8.hip — one workgroup per 16-column tile, waves own contiguous K slices of up to TPW tiles of
// BOTH planes, everything requested up front, the activation rows staged per wave into a wave-private LDS strip, one
// barrier, bias in the epilogue. Scope: per-128 groups or one group per column (scale_mode 0) and, round 5, groups of
// 32 / 64 / 96 (scale_mode 1: the reference's DEFAULT group size is 32, utils/config.py:794-842), unshuffled aligned rows, K up
// to 8192 as four tiles per wave x up to sixteen waves, K up to 12288 as eight tiles per wave x up to twelve waves
// (round 5: the 7B down_proj, K = 11008, with its parity test); anything else keeps the lookup kernel.
#include <algorithm>
#include <cstdlib>

#include "woq_gemv_common.h"

namespace woq {

constexpr int F8_DIG = 6;   // base-16 digits per activation value
constexpr int F8_SETM = 2;  // activation rows per MFMA row set (12 of the 16 MFMA rows)
constexpr int F8_MAXM = 8;

__host__ __device__ constexpr int f8_row_bytes(int TPW) { return TPW * 128 + 16; }
// LDS: [nw zero blocks of 256][nw strips: F8_DIG * ms digit rows x row bytes][slab nrs x nw x F8_SETM x 16 f32]
// (+ per-32 scales: one all-zero strip row per wave, so that masked A operands keep the live rows' constant offsets)
__host__ __device__ inline size_t f8_lds_bytes(int M, int ms, int nw, int TPW, int smode = 0) {
  const int nrs = (M + ms - 1) / ms;
  return (size_t)nw * 256 + (size_t)nw * F8_DIG * ms * f8_row_bytes(TPW) + (size_t)nrs * nw * F8_SETM * 16 * 4 +
         (smode ? (size_t)nw * f8_row_bytes(TPW) : 0);
}

typedef long i64_t;

template <bool E5M2>
__device__ __forceinline__ float4_t mfma_f8(i64_t a, i64_t b, float4_t c) {
  if constexpr (E5M2)
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a, b, c, 0, 0, 0);  // A: e4m3 digits, B: e5m2 codes
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0);
}

// flags: bit 0 scales are bf16 (else fp16; ignored for fp32 scales), bits 2-3 activation rows 0 fp32 | 1 fp16 | 2 bf16
// TPW = 4: up to sixteen waves (K <= 8192); TPW = 8 (round 5): up to twelve waves of eight tiles of both planes
// (K <= 12288: a 7B down_proj's K = 11008 is eleven waves) — 128 weight registers per lane, hence the 768-thread bound
// SMODE 1 (round 5): groups of 32 / 64 / 96 — the blob carries one scale per 32-k block (include/woq_blob.h). A 32-k MFMA
// of this kernel contracts over all four lane quarters of a 64-k half, i.e. over BOTH of its 32-k blocks, so each half is
// issued twice with the A operand of the other block's two quarters read from the zero block (the int4 tile kernel's form
// for per-32 scales at M > 1): two more MFMAs and one more recombination per half, the matrix pipe has the room.
// Round 6 (fp8-weight layers inside the decode engine): the epilogues the engine's launches carry. norm_w (fp32 [K],
// fp32 activation rows only): HF LlamaRMSNorm folded in — the rows are staged as x * norm_w (every wave sums the squares
// of the RAW x of its slice on the way) and the result is multiplied by rsqrt(mean(x^2) + eps) in the epilogue, where
// the scale factors already sit (the factor is one number per row: it commutes with the inner product). residual (fp32
// [M][ld_res]): out = residual + result (may alias out: a column is read and written by the same thread).
struct F8Fused {
  const float* norm_w;
  float eps;
  const float* residual;
  int ld_res;
};

template <int TPW, bool E5M2, bool S32, int SMODE = 0>
__global__ __launch_bounds__(TPW == 8 ? 768 : 1024) void gemv_fp8_kernel(
    const u32x4* __restrict__ qhi, const u32x4* __restrict__ qlo, const void* __restrict__ scales,
    const void* __restrict__ x, int tiles_k, int K, int base_tiles, int rem_tiles, int n_groups, int tpg_shift,
    void* __restrict__ out, const float* __restrict__ bias, int N, int M, int ms, int lda, int ldo, int out_dtype,
    int flags, F8Fused fz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float ssq_l[F8_MAXM][16];  // RMSNorm: partial sums of squares of row m's slice, per wave
  constexpr int RB = f8_row_bytes(TPW);
  constexpr int XJ = TPW / 2;  // float4 loads per lane per row covering TPW * 128 activations
  constexpr int ESZ = S32 ? 4 : 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int nrs = (M + ms - 1) / ms;
  unsigned char* zero_blk = smem_raw + wid * 256;
  unsigned char* strips = smem_raw + nw * 256;
  unsigned char* strip = strips + (size_t)wid * F8_DIG * ms * RB;       // this wave's [F8_DIG * ms][RB]
  float* slab = (float*)(strips + (size_t)nw * F8_DIG * ms * RB);      // [nrs][nw][F8_SETM][16]
  const int kt0 = wid * base_tiles + min(wid, rem_tiles);
  const int cnt = base_tiles + (wid < rem_tiles ? 1 : 0);
  const int i16 = lane & 15, kq = lane >> 4;
  const int kbase = kt0 * 128;
  const int xlen = max(0, min(cnt * 128, K - kbase));
  const bool bf = (flags & 1) != 0;
  const int xdt = (flags >> 2) & 3;
  const int v16 = lane * 16;
  const int tn = (int)blockIdx.x;

  // ---- 1. both planes of the wave's K slice, and its group scales (everything past the slice reads as zero) ----
  u32x4 wh[TPW], wl[TPW];
  {
    const int qbytes = min(kt0 + cnt, tiles_k) * 1024;
    const rsrc_t rh = make_rsrc(qhi + (size_t)tn * tiles_k * 64, qbytes);
    const rsrc_t rl = make_rsrc(qlo + (size_t)tn * tiles_k * 64, qbytes);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      wh[t] = __builtin_amdgcn_raw_buffer_load_b128(rh, v16 + t * 1024, kt0 * 1024, AUX_NT);
      wl[t] = __builtin_amdgcn_raw_buffer_load_b128(rl, v16 + t * 1024, kt0 * 1024, AUX_NT);
    }
  }
  typename RawSc<SMODE, S32>::type rsc[TPW];
  if constexpr (SMODE == 0) {
    const rsrc_t rs = make_rsrc((const char*)scales + (size_t)tn * n_groups * 16 * ESZ, n_groups * 16 * ESZ);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int grp = min((kt0 + t) >> tpg_shift, n_groups - 1);
      if constexpr (S32)
        rsc[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i16 * 4, grp * 64, 0));
      else
        rsc[t] = __builtin_amdgcn_raw_buffer_load_b16(rs, i16 * 2, grp * 32, 0);
    }
  } else {  // four scales per (tile, column): [tn][kt][16][4]
    const rsrc_t rs = make_rsrc((const char*)scales + (size_t)tn * tiles_k * 64 * ESZ, tiles_k * 64 * ESZ);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if constexpr (S32)
        rsc[t] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, i16 * 16 + t * 256, kt0 * 256, 0));
      else
        rsc[t] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, i16 * 8 + t * 128, kt0 * 128, 0));
    }
  }
  ((uint32_t*)zero_blk)[lane] = 0u;
  unsigned char* zrow = nullptr;  // per-32 scales: this wave's all-zero strip row (behind the slab)
  if constexpr (SMODE == 1) {
    zrow = (unsigned char*)(slab + (size_t)nrs * nw * F8_SETM * 16) + (size_t)wid * RB;
    for (int j = lane; j < RB / 4; j += 64) ((uint32_t*)zrow)[j] = 0u;
  }

  auto load_row = [&](size_t row_off, float4_t (&xv)[XJ], int m_abs) {
    if (xdt == 0) {
      const rsrc_t rx = make_rsrc((const float*)x + row_off + kbase, xlen * 4);
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        xv[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, v16 + j * 1024, 0, 0));
      if (fz.norm_w != nullptr) {
        const rsrc_t rg = make_rsrc(fz.norm_w + kbase, xlen * 4);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
          const float4_t g4 = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, v16 + j * 1024, 0, 0));
          ss = fmaf(xv[j].x, xv[j].x, fmaf(xv[j].y, xv[j].y, fmaf(xv[j].z, xv[j].z, fmaf(xv[j].w, xv[j].w, ss))));
          xv[j] = xv[j] * g4;
        }
        ss = wave_sum_dpp(ss);
        if (lane == 0) ssq_l[m_abs][wid] = ss;
      }
    } else {
      const rsrc_t rx = make_rsrc((const uint16_t*)x + row_off + kbase, xlen * 2);
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const uint2 r = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rx, lane * 8 + j * 512, 0, 0));
        const uint16_t hb[4] = {(uint16_t)r.x, (uint16_t)(r.x >> 16), (uint16_t)r.y, (uint16_t)(r.y >> 16)};
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float fb = bf16_bits_to_f32(hb[i]), fh = f16_bits_to_f32(hb[i]);
          f[i] = xdt == 2 ? fb : fh;
        }
        xv[j] = (float4_t){f[0], f[1], f[2], f[3]};
      }
    }
  };

  // row m of the set -> six digit rows of the strip; returns 2^(e - 21), the factor that undoes the fixed point
  auto stage_row = [&](int m, float4_t (&xv)[XJ]) -> float {
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < XJ; ++j)
      amax = fmaxf(fmaxf(amax, fabsf(xv[j].x)), fmaxf(fabsf(xv[j].y), fmaxf(fabsf(xv[j].z), fabsf(xv[j].w))));
    amax = wave_max_dpp(amax);
    int e = 0;
    if (amax > 0.f && amax < INFINITY) e = max(-100, min(100, __builtin_amdgcn_frexp_expf(amax)));
    const float sfix = ldexpf(1.f, 21 - e);
    unsigned char* r0 = strip + (size_t)(F8_DIG * m) * RB + lane * 4;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      // amax < 2^e  =>  |x| * 2^(21 - e) < 2^21: the fp32 sum with 1.5 * 2^23 has ulp 1, its mantissa IS the rounded
      // integer + 2^22 (woq_gemv_i8.hip stage_row)
      int v[4] = {(int)(__float_as_uint(fmaf(xv[j].x, sfix, 12582912.f)) & 0x7fffffu) - (1 << 22),
                  (int)(__float_as_uint(fmaf(xv[j].y, sfix, 12582912.f)) & 0x7fffffu) - (1 << 22),
                  (int)(__float_as_uint(fmaf(xv[j].z, sfix, 12582912.f)) & 0x7fffffu) - (1 << 22),
                  (int)(__float_as_uint(fmaf(xv[j].w, sfix, 12582912.f)) & 0x7fffffu) - (1 << 22)};
#pragma unroll
      for (int dg = 0; dg < F8_DIG; ++dg) {
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = (int)((uint32_t)v[i] << 28) >> 28;  // low nibble, sign-extended: [-8, 7]
          v[i] = (v[i] - d) >> 4;
          f[i] = (float)d;
        }
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);  // exact: integers up to 16 are e4m3 values
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
        *(uint32_t*)(r0 + (size_t)dg * RB + j * 256) = (uint32_t)w;
      }
    }
    return ldexpf(1.f, e - 21);
  };

#pragma unroll 1
  for (int rs = 0; rs < nrs; ++rs) {  // row sets over the same weight registers
    const int Mrs = min(ms, M - rs * ms);
    __builtin_amdgcn_wave_barrier();
    float unsc[F8_SETM] = {0.f, 0.f};
    for (int m = 0; m < Mrs; ++m) {
      float4_t xv[XJ];
      load_row((size_t)(rs * ms + m) * lda, xv, rs * ms + m);
      const float u = stage_row(m, xv);
      if (m == 0)
        unsc[0] = u;
      else
        unsc[1] = u;
    }
    // the strip and the zero block are wave-private and LDS executes one wave's accesses in order: no workgroup barrier
    __builtin_amdgcn_wave_barrier();

    // A rows: MFMA row r = lane & 15 -> activation row r / 6 of the set, digit r % 6; rows past 6 * Mrs read zeros.
    // D: lane quarter kq holds rows 4 kq .. 4 kq + 3 of column lane & 15.
    const int a_m = i16 / F8_DIG, a_dg = i16 - a_m * F8_DIG;
    const bool a_live = a_m < Mrs;
    const unsigned char* a_base = a_live ? strip + (size_t)(F8_DIG * a_m + a_dg) * RB + kq * 16 : zero_blk + kq * 16;
    const int st_t = a_live ? 128 : 0, st_h = a_live ? 64 : 0, st_s = a_live ? 8 : 0;
    const unsigned char* a_row = strip + (size_t)(F8_DIG * a_m + a_dg) * RB + kq * 16;
    const unsigned char* a_blk0 = (SMODE == 1 && a_live && kq < 2) ? a_row : zrow + kq * 16;
    const unsigned char* a_blk1 = (SMODE == 1 && a_live && kq >= 2) ? a_row : zrow + kq * 16;
    // this lane's four result rows as (activation row, digit): weights 16^digit of the recombination
    float c0[4], c1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * kq + i, m = r / F8_DIG, dg = r - m * F8_DIG;
      const float p = ldexpf(1.f, 4 * dg);
      c0[i] = (m == 0 && Mrs > 0) ? p : 0.f;
      c1[i] = (m == 1 && Mrs > 1) ? p : 0.f;
    }
    float tot0 = 0.f, tot1 = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      // (per row set the code bytes are rebuilt: opaque to the optimiser, or it keeps all of them live across the loop)
      asm volatile("" : "+v"(wh[t]), "+v"(wl[t]));
      float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t h0 = h == 0 ? wh[t].x : wh[t].z, h1 = h == 0 ? wh[t].y : wh[t].w;
        const uint32_t l0 = h == 0 ? wl[t].x : wl[t].z, l1 = h == 0 ? wl[t].y : wl[t].w;
        // code = (hi nibble << 4) | (lo nibble ^ 8); k order of a dword pair: low nibbles, high nibbles (woq_blob.h)
        const uint32_t b0 = ((h0 & 0x0f0f0f0fu) << 4) | ((l0 & 0x0f0f0f0fu) ^ 0x08080808u);
        const uint32_t b1 = (h0 & 0xf0f0f0f0u) | (((l0 >> 4) & 0x0f0f0f0fu) ^ 0x08080808u);
        const uint32_t b2 = ((h1 & 0x0f0f0f0fu) << 4) | ((l1 & 0x0f0f0f0fu) ^ 0x08080808u);
        const uint32_t b3 = (h1 & 0xf0f0f0f0u) | (((l1 >> 4) & 0x0f0f0f0fu) ^ 0x08080808u);
        const i64_t bq0 = (i64_t)(((unsigned long)b1 << 32) | b0), bq1 = (i64_t)(((unsigned long)b3 << 32) | b2);
        if constexpr (SMODE == 0) {
          const i64_t a_lo = *(const i64_t*)(a_base + t * st_t + h * st_h);
          const i64_t a_hi = *(const i64_t*)(a_base + t * st_t + h * st_h + st_s);
          acc = mfma_f8<E5M2>(a_lo, bq0, acc);
          acc = mfma_f8<E5M2>(a_hi, bq1, acc);
        } else {
          // 32-k block g of this half = lane quarters 2 g, 2 g + 1: the other two quarters' A rows read as zeros
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            // live rows of the block's own quarters read the strip, everything else the zero row — at the SAME constant
            // offsets, so the addresses are one base register per block + immediates (per-lane strides cost a VGPR per
            // address: 130 spilled registers in the first form)
            const unsigned char* ab = g == 0 ? a_blk0 : a_blk1;
            float4_t ag = {0.f, 0.f, 0.f, 0.f};
            ag = mfma_f8<E5M2>(*(const i64_t*)(ab + t * 128 + h * 64), bq0, ag);
            ag = mfma_f8<E5M2>(*(const i64_t*)(ab + t * 128 + h * 64 + 8), bq1, ag);
            float sg;
            if constexpr (S32) {
              const float4_t r4 = rsc[t];
              sg = h == 0 ? (g == 0 ? r4.x : r4.y) : (g == 0 ? r4.z : r4.w);
            } else {
              const uint32_t r = h == 0 ? rsc[t].x : rsc[t].y;
              sg = tscale16(g == 0 ? (r & 0xffffu) : (r >> 16), bf);
            }
            tot0 = fmaf(sg, fmaf(ag.x, c0[0], fmaf(ag.y, c0[1], fmaf(ag.z, c0[2], ag.w * c0[3]))), tot0);
            tot1 = fmaf(sg, fmaf(ag.x, c1[0], fmaf(ag.y, c1[1], fmaf(ag.z, c1[2], ag.w * c1[3]))), tot1);
          }
        }
      }
      // (per-32 scales: left alone, the scheduler hoists the A-operand reads of every tile — 16 per tile — above the
      // loop and spills; one tile's reads at a time)
      if constexpr (SMODE == 1) __builtin_amdgcn_sched_barrier(0);
      if constexpr (SMODE == 0) {
        float sc;
        if constexpr (S32)
          sc = rsc[t];
        else
          sc = tscale16(rsc[t], bf);
        tot0 = fmaf(sc, fmaf(acc.x, c0[0], fmaf(acc.y, c0[1], fmaf(acc.z, c0[2], acc.w * c0[3]))), tot0);
        tot1 = fmaf(sc, fmaf(acc.x, c1[0], fmaf(acc.y, c1[1], fmaf(acc.z, c1[2], acc.w * c1[3]))), tot1);
      }
    }
    // a row's six digit sums sit in two lane quarters: sum the four quarters (the unused ones hold zeros)
    tot0 = reduce_kq(tot0) * unsc[0];
    tot1 = reduce_kq(tot1) * unsc[1];
    if (lane < 16) {
      float* s = slab + (((size_t)rs * nw + wid) * F8_SETM) * 16 + lane;
      s[0] = tot0;
      s[16] = tot1;
    }
  }
  __syncthreads();

  // ---- finish: sum over waves, bias, store ----
  for (int idx = tid; idx < M * 16; idx += (int)blockDim.x) {
    const int e_i = idx & 15, e_m = idx >> 4;
    const int e_rs = e_m / ms, e_mi = e_m - e_rs * ms;
    float v = 0.f;
    for (int w2 = 0; w2 < nw; ++w2) v += slab[(((size_t)e_rs * nw + w2) * F8_SETM + e_mi) * 16 + e_i];
    const int n = tn * 16 + e_i;
    if (n < N) {
      if (fz.norm_w != nullptr) {
        float ss = 0.f;
        for (int w2 = 0; w2 < nw; ++w2) ss += ssq_l[e_m][w2];
        v *= 1.0f / sqrtf(ss / (float)K + fz.eps);  // HF LlamaRMSNorm
      }
      if (bias) v += bias[n];
      if (fz.residual != nullptr) v += fz.residual[(size_t)e_m * fz.ld_res + n];
      store_f32(out, (size_t)e_m * ldo + n, out_dtype, v);
    }
  }
}

struct F8Launch {
  const void *qhi, *qlo, *scales, *x;
  int tiles_k, K, N, n_groups, tpg_shift, M, ms, lda, ldo, out_dtype, flags, nw, grid;
  void* out;
  const float* bias;
  F8Fused fz;
};

template <int TPW, bool E5M2, bool S32, int SMODE>
static int launch_fp8_t(const F8Launch& a, hipStream_t st) {
  const size_t lds = f8_lds_bytes(a.M, a.ms, a.nw, TPW, SMODE);
  auto kern = gemv_fp8_kernel<TPW, E5M2, S32, SMODE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       158 * 1024);  // the kernel also holds 512 B of static LDS (the RMSNorm partials)
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  const int base = a.tiles_k / a.nw, rem = a.tiles_k % a.nw;
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(a.nw * 64), lds, st, (const u32x4*)a.qhi, (const u32x4*)a.qlo, a.scales,
                     a.x, a.tiles_k, a.K, base, rem, a.n_groups, a.tpg_shift, a.out, a.bias, a.N, a.M, a.ms, a.lda,
                     a.ldo, a.out_dtype, a.flags, a.fz);
  return 0;
}

// geometry: 4 tiles per wave and up to 16 waves (K <= 8192), else 8 tiles per wave and up to 12 waves (K <= 12288)
static bool fp8_geometry(int tiles_k, int& nw, int& tpw) {
  tpw = tiles_k > 64 ? 8 : 4;
  nw = (tiles_k + tpw - 1) / tpw;
  return nw >= 1 && nw <= (tpw == 8 ? 12 : 16);
}

// Does the fp8-MFMA kernel take this call? `hi` = the HI plane's header (scales; the LO plane has the same geometry).
bool gemv_fp8_mfma_supported(const void* act, int act_dtype, int lda, const woq_blob_header& hi) {
  static const bool off = getenv("WOQ_FP8_GENERIC") != nullptr;  // A/B switch: the lookup kernel
  if (off || hi.off_shuffle != 0 || hi.off_zp != 0 || hi.scale_mode > 1 || (hi.K & 3) != 0 || (lda & 3) != 0 ||
      (((uintptr_t)act) & (act_dtype == WOQ_F32 ? 15 : 7)) != 0)
    return false;
  // scale_mode 0: per-128 groups or one group per column; scale_mode 1 (round 5): one scale per 32-k block (groups 32 / 64 / 96)
  if (hi.scale_mode == 0 && hi.n_groups > 1 && hi.group != WOQ_TILE_K) return false;
  int nw, tpw;
  return fp8_geometry(hi.Kpad / WOQ_TILE_K, nw, tpw);
}

// rows 0..M-1 (M <= 8) of an fp8 weight: act [M, lda], out [M, ldo]
int launch_gemv_fp8_mfma(const void* act, int act_dtype, int lda, int M, const void* hi_blob, const woq_blob_header& hi,
                         const void* lo_q, uint32_t fp8_type, const float* bias, void* out, int out_dtype, int ldo,
                         hipStream_t st, const float* norm_w, float eps, const float* residual, int ld_res) {
  F8Launch a;
  if (norm_w != nullptr && act_dtype != WOQ_F32) return woq::fail("QBits: the fp8 GEMV's fused RMSNorm takes fp32 rows");
  a.fz = F8Fused{norm_w, eps, residual, ld_res};
  const uint8_t* b = (const uint8_t*)hi_blob;
  a.qhi = b + hi.off_q;
  a.qlo = lo_q;
  a.scales = b + hi.off_scale;
  a.x = act;
  a.tiles_k = hi.Kpad / WOQ_TILE_K;
  a.K = hi.K;
  a.N = hi.N;
  a.n_groups = hi.n_groups;
  a.tpg_shift = 0;  // one tile per group (or one group): gemv_fp8_mfma_supported
  a.M = M;
  a.lda = lda;
  a.ldo = ldo;
  a.out = out;
  a.out_dtype = out_dtype;
  a.bias = bias;
  a.flags = (hi.scale_type == WOQ_BF16 ? 1 : 0) | (act_dtype == WOQ_F16 ? 4 : (act_dtype == WOQ_BF16 ? 8 : 0));
  int tpw;
  if (M < 1 || M > F8_MAXM || !fp8_geometry(a.tiles_k, a.nw, tpw))
    return woq::fail("QBits: shape not covered by the fp8 decode GEMV");
  a.ms = std::min(M, F8_SETM);
  const int sm = (int)hi.scale_mode;
  if (f8_lds_bytes(M, a.ms, a.nw, tpw, sm) > 150 * 1024) a.ms = 1;  // long K: one activation row per set
  if (f8_lds_bytes(M, a.ms, a.nw, tpw, sm) > 150 * 1024) return woq::fail("QBits: activation rows do not fit LDS");
  a.grid = hi.Npad / WOQ_TILE_N;
  const bool e5m2 = fp8_type == WOQ_W_FP8_E5M2, s32 = hi.scale_type == WOQ_F32;
const int smode = (int)hi.scale_mode;
#define WOQ_F8_CASE(T, E, S)                                                     \
  if (tpw == T && e5m2 == E && s32 == S)                                         \
    return smode == 0 ? launch_fp8_t<T, E, S, 0>(a, st) : launch_fp8_t<T, E, S, 1>(a, st);
  WOQ_F8_CASE(4, false, false)
  WOQ_F8_CASE(4, false, true)
  WOQ_F8_CASE(4, true, false)
  WOQ_F8_CASE(4, true, true)
  WOQ_F8_CASE(8, false, false)
  WOQ_F8_CASE(8, false, true)
  WOQ_F8_CASE(8, true, false)
  WOQ_F8_CASE(8, true, true)
#undef WOQ_F8_CASE
  return woq::fail("QBits: bad fp8 GEMV configuration");
}

// act[p * 16 + i] = SiLU(gu[(2 p) * 16 + i]) * gu[(2 p + 1) * 16 + i]: the fused gate/up projection's 16-column tiles
// alternate gate / up (include/woq_hip.h woq_layer_weights) — the fp8 GEMV writes them as they come, this pairs them
__global__ __launch_bounds__(256) void silu_mul_tiles_kernel(const float* __restrict__ gu, int inter, float* __restrict__ act) {
  const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (i < inter) {
    const int p = i >> 4, c = i & 15;
    const float g = gu[(size_t)(2 * p) * 16 + c], u = gu[(size_t)(2 * p + 1) * 16 + c];
    act[i] = g / (1.0f + __expf(-g)) * u;
  }
}
void launch_silu_mul_tiles(const float* gu, int inter, float* act, hipStream_t st) {
  hipLaunchKernelGGL(silu_mul_tiles_kernel, dim3((inter + 255) / 256), dim3(256), 0, st, gu, inter, act);
}

}  // namespace woq
