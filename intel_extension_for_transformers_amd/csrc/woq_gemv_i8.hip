// woq_gemv_i8.hip — small-M (1..16 rows, four per MFMA row set) int4 GEMV, the per-token hot kernel of the fp32-activation path.
//
// Arithmetic and parity definition (reference): qbits.cpp:113-140 -> bestla_weightonly_dispatcher.cpp:120-189
// (per N-tile x K-block: unpack int4, apply scale / zero point, fp32 accumulate, epilogue
// alpha*acc + beta*bias, bestla_customop.hpp:22-40); definition autograd/functions.py:41-63.
//
// What bounds it. A batch-1 projection is 8-45 MB of weights read exactly once: 1-6 us at the HBM roof.
// Measured on MI355X (profiles/r01*_gemv_probe_*.txt): a CU's vector-memory path sustains ~10 B/clk of HBM
// misses, i.e. one 1-KiB weight tile per ~100 clk per CU or ~400 clk per SIMD, so a kernel that spends more
// than ~100 wave-instructions per tile (all fixed costs included) is ISSUE-bound, not HBM-bound. The first
// two generations of this kernel (fp32 VALU: ~150 instr/tile; fp16 MFMA with magic-number dequantisation:
// ~110) were exactly that. This one spends ~30:
//  * integer inner product on the matrix pipe, v_mfma_i32_16x16x64_i8. The blob stores every int4 as a signed
//    nibble in the MFMA's B-fragment order (include/woq_blob.h), so `(w << 4) & 0xf0f0f0f0` and
//    `w & 0xf0f0f0f0` ARE four int8 values 16*q each: 3 VALU per 8 weights, 2 MFMAs per 128-k tile;
//  * the activation row is converted ONCE per wave slice to 22-bit+sign offset-binary fixed point relative to the
//    slice maximum (v*2^s + 1.5*2^23 in fp32: the mantissa bits of the sum ARE the integer), whose three
//    bytes are three int8 "limb" rows of the A operand (rows 4m..4m+2 of activation row m; row 4m+3 is all
//    ones, so the MFMA also returns sum_k q_k, which removes the limb biases exactly). All products and sums
//    are exact in int32; per tile the three limb sums are recombined in fp32 and scaled by the group scale.
//    Net: the weight side is exact, the activation carries |x - x~| <= max|x_slice| * 2^-22 — the absolute
//    accuracy of an fp32 product — and results do not depend on summation order inside a tile;
//  * zero points (asym): sum_k (q_k - zp) x_k = sum_k q_k x_k - zp * sum_k x_k, with sum_k x~_k per tile from
//    one extra MFMA pair against an all-ones B (the same identity BesTLA uses, dispatcher.cpp:154-160);
//  * per-32 scales (group 32 / 64): at batch 1 the MFMA's output rows 4..7 are free, so they carry the same
//    activation row restricted to the second 32-k group of each 64-k block (rows 0..3: the first group) — ONE MFMA
//    per 64-k half returns both groups' sums, each lane quarter applies its own group's scale and zero point;
//    for 2..4 rows the 64-k MFMA is issued twice with the A operand of the other 32-k half read from a zero block.
// Schedule:
//  * one workgroup = CB adjacent 16-column tiles (CB = 2 for the fused gate/up SiLU*mul pairs) x all of K;
//    its waves own contiguous balanced K slices of up to TPW = 8 tiles (8 KiB per column tile) each;
//  * a wave issues its activation-row loads first (they come back from L2 first: loads return in order), then
//    all scale / zero-point loads and ALL of its weight tiles (buffer_load_dwordx4 nt, straight to VGPRs): the
//    CU's miss queue holds ~16 KiB, so the wave may wait at issue, but queueing everything before the staging
//    measured +2.3 % tokens/s over keeping only four tiles in flight (WOQ_PF below); every wait is a counted vmcnt;
//  * while the weights fly, the wave stages ONLY ITS OWN K slice of the activation rows into a wave-private
//    LDS strip — no workgroup barrier, no full-vector dependency: RMSNorm is separable
//    (out = rsqrt(mean(x^2)+eps) * (W . (x*g)): every wave adds its slice's sum of squares to the reduction
//    slab, the factor is applied in the epilogue) and the fixed-point scale is per wave slice;
//  * buffer descriptors with exact bounds replace every clamp / mask (out-of-range reads return 0);
//  * ONE barrier per workgroup: partial sums -> LDS slab -> the first CB*M*16 threads finish
//    (RMSNorm factor, bias, residual, SiLU*mul) and store.
// Kernel arguments are plain scalars (not a struct) so the first 14 dwords are preloaded into SGPRs at dispatch
// (-mllvm -amdgpu-kernarg-preload-count, see the Makefile): no s_load round trip before the first address.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "woq_gemv_common.h"

// Probe hooks (tools/gemv_probe.hip compiles this file with WOQ_PROBE, plus WOQ_PROBE_STAMPS for the per-stage
// timeline — the stamps cost ~300 cycles each, so timings are taken without them). Nothing in the product build.
#ifdef WOQ_PROBE
extern int g_probe_flags;
#define WOQ_SKIP(bit) ((flags >> (bit)) & 1)  // experiment switches (bits 4..9 of flags)
#else
#define WOQ_SKIP(bit) false
#endif
#ifdef WOQ_PROBE_STAMPS
extern __device__ unsigned long long* g_probe;
#define WOQ_STAMP(k)                                                              \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    if (g_probe && lane == 0) {                                                   \
      unsigned long long* slot_ = g_probe + ((size_t)blockIdx.x * 16 + wid) * 32; \
      slot_[(k)] = clock64();                                                     \
      if ((k) == 0) slot_[31] = wall_clock64();                                   \
    }                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)
#else
#define WOQ_STAMP(k)
#endif

// weight-tile loads issued before the activation staging (the rest follow one per consumed tile, or right after the
// staging with WOQ_REST_EARLY). Measured on the Llama-2-7B decode (tokens/s): 4 -> 721, 8 -> 729, 16 = everything up
// front -> 740, 8 + rest-early -> 744 (= 16 for every projection but gate/up), 4 + rest-early -> 719: the earlier
// the whole wave's loads are queued the better, even when the CU's miss queue makes the wave wait at issue.
#ifndef WOQ_PF
#define WOQ_PF 16
#endif
#ifndef WOQ_REST_EARLY
#define WOQ_REST_EARLY 0
#endif
namespace woq {


constexpr int TSETM = 4;  // activation rows per MFMA row set: 4 MFMA rows (3 limbs + ones) per activation row
constexpr int TMAXM = 16;  // activation rows per launch: up to four row sets over the SAME weight registers (round 3:
                           // rows 5..8 used to be a second launch that streamed the weights again, 17-21 us against
                           // 10.9 at M = 4 for the qkv shape — profiles/r03p; rows 9..16 used to go to the MFMA GEMM)


// LDS (dynamic, bytes): [nw zero blocks of 256][nw ones blocks of 256][nw strips: 3*min(M,4) limb rows x (TPW*128 + 16)]
//                       [slab nrs x nw x CB x 64 f32][sumsq nrs x nw x 4 f32], nrs = row sets = ceil(M / 4)
__host__ __device__ constexpr int tile_row_bytes(int TPW) { return TPW * 128 + 16; }  // +16 B: rows on distinct banks
__host__ __device__ inline size_t tile_lds_bytes(int M, int nw, int TPW, int CB) {
  const int ms = M < TSETM ? M : TSETM, nrs = (M + TSETM - 1) / TSETM;
  return (size_t)nw * 512 + (size_t)nw * 3 * ms * tile_row_bytes(TPW) + (size_t)nrs * nw * CB * 64 * 4 +
         (size_t)nrs * nw * TSETM * 4;
}



// flags: bit 0 scales are bf16 (else fp16; ignored for fp32 scales), bit 1 SiLU(gate)*up epilogue (CB == 2)
// NDIG: 0 = int4 weights; 1 | 2 | 3 = a 4-bit table type (nf4 / fp4) as that many digit planes (woq_gemv_common.h LutArgs)
// SHUF (round 5, batch 1, int4): GPTQ act-order blobs — weight row k meets activation x[shuffle[k]] (the converted
// g_idx the blob carries; reference semantics: `index_select(x, 1, g_idx)` in autograd/functions.py:41-63 and BesTLA's
// ShuffleActivation prologue, bestla_weightonly_dispatcher.cpp:138-142,163-166). The workgroup first copies the WHOLE
// activation vector (times the RMSNorm weight) into LDS with coalesced requests — a first form gathered straight from
// memory, 32 four-byte requests per lane hitting 64 different lines each: 12.8 us per launch against 7.4 without the
// gather (profiles/r05d_*) — then every wave picks its slice's elements out of LDS by the blob's index vector and stages
// them exactly like a contiguous slice. One extra workgroup barrier, under the weight stream. Everything else is the
// kernel above.
template <int TPW, int CB, int SMODE, bool ASYM, bool S32, bool M1, int NDIG, bool SHUF = false>
// register budget by workgroup size: 1024 threads -> 128 VGPRs (group-128 paths), 768 -> 168 (per-32 scales keep
// 3 more registers per tile and twice the A fragments), 512 -> 256
__global__ __launch_bounds__(CB * TPW > 8 ? 512 : (SMODE == 1 ? 768 : 1024)) void gemv_tile_kernel(
    const u32x4* __restrict__ q, const void* __restrict__ scales, const void* __restrict__ x,
    const float* __restrict__ norm_w, int tiles_k, int K, int base_tiles, int rem_tiles, int n_groups, int tpg_shift,
    const uint8_t* __restrict__ zp, void* __restrict__ out, const float* __restrict__ bias, const float* residual,
    float eps, int N, int Mrows, int lda, int ldo, int ld_res, int out_dtype, int flags, int kt_off, LutArgs lut,
    const int32_t* __restrict__ shuffle) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  static_assert(!(ASYM && NDIG > 0), "table weight types are symmetric");
  static_assert(!SHUF || (M1 && NDIG == 0), "the act-order gather is built for batch-1 int4 launches");
  constexpr int NB = NDIG > 0 ? NDIG : 1;  // B operands per 64-k half
  // one 64-k half against the A operand `av`: sum_k (activation) x (16 q | table digits), recombined in fp32
  auto half_dot = [&](const i32x4& av, uint32_t w0, uint32_t w1) -> float {
    i32x4 b[NB];
    if constexpr (NDIG > 0)
      lut_b<NDIG>(lut, w0, w1, b);
    else
      int4_b(w0, w1, b[0]);
    float f = limb_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[NB - 1], i32x4{0, 0, 0, 0}, 0, 0, 0));
#pragma unroll
    for (int j = NB - 2; j >= 0; --j)
      f = fmaf(f, 256.f, limb_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[j], i32x4{0, 0, 0, 0}, 0, 0, 0)));
    return f;
  };
  constexpr int RB = tile_row_bytes(TPW);
  constexpr int XJ = TPW / 2;  // float4 loads per lane per row covering TPW*128 activations
  constexpr int ESZ = S32 ? 4 : 2;
  static_assert((TPW & 1) == 0, "TPW must be even");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int M = M1 ? 1 : Mrows;
  const int MS = M1 ? 1 : min(M, TSETM);          // rows of one set (the strip holds one set at a time)
  const int nrs = M1 ? 1 : (M + TSETM - 1) / TSETM;  // row sets: each runs over the same weight registers
  unsigned char* zero_blk = smem_raw + wid * 256;             // this wave's 256 B of zeros
  unsigned char* ones_blk = smem_raw + nw * 256 + wid * 256;  // this wave's 256 B of int8 ones
  unsigned char* strips = smem_raw + nw * 512;
  unsigned char* strip = strips + (size_t)wid * 3 * MS * RB;  // this wave's [3*MS][RB]
  float* slab = (float*)(strips + (size_t)nw * 3 * MS * RB);  // [nrs][nw][CB][64]
  float* ssq = slab + (size_t)nrs * nw * CB * 64;              // [nrs][nw][TSETM]

  // this wave's K tiles: balanced contiguous slice [kt0, kt0 + cnt), cnt <= TPW. Everything past the slice end
  // reads as zero through the descriptors' bounds, so tiles t >= cnt contribute exactly 0.
  // kt_off: first K tile of this launch (K ranges beyond one launch's reach are covered by chained launches)
  const int kt0 = kt_off + wid * base_tiles + min(wid, rem_tiles);
  const int cnt = base_tiles + (wid < rem_tiles ? 1 : 0);
  const int i16 = lane & 15, kq = lane >> 4;
  const int kbase = kt0 * 128;
  const int xlen = max(0, min(cnt * 128, K - kbase));  // activations of this slice
  const bool norm = norm_w != nullptr;
  const bool bf = (flags & 1) != 0, silu = (flags & 2) != 0;
  const int v16 = lane * 16;
  WOQ_STAMP(0);

  // the thread's residual element (batch-1 form), fetched up front through a descriptor that is empty when
  // there is no residual: no branch, and the epilogue does not end on a dependent global load
  float e_res = 0.f;
  if constexpr (M1) {
    const int n0 = silu ? (int)blockIdx.x * 16 : (int)blockIdx.x * CB * 16;
    const int nlim = silu ? (N >> 1) : N;
    const rsrc_t rr = make_rsrc(residual ? (const void*)(residual + n0) : x,
                                residual ? max(0, min(nlim - n0, CB * 16)) * 4 : 0);
    e_res = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, min(tid, 63) * 4, 0, 0));
  }

  // ---- 0. row 0 of the activations (and the RMSNorm weight) first ----
  // The activation rows are fp32 (what the reference's boundary holds, modules.py:152-154) or — read natively, every
  // fp16 / bf16 value is an exact fp32 — 16-bit (flags bits 2-3: 1 fp16, 2 bf16): 8-byte loads of the same four
  // elements per lane, widened in registers.
  const int xdt = (flags >> 2) & 3;
  auto load_row = [&](size_t row_off, float4_t (&xv)[XJ]) {
    if (xdt == 0) {
      const rsrc_t rx = make_rsrc((const float*)x + row_off + kbase, WOQ_SKIP(7) ? 0 : xlen * 4);
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        xv[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, v16 + j * 1024, 0, 0));
    } else {
      const rsrc_t rx = make_rsrc((const uint16_t*)x + row_off + kbase, WOQ_SKIP(7) ? 0 : xlen * 2);
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
  const rsrc_t rg = make_rsrc(norm ? (const void*)(norm_w + kbase) : x, norm ? xlen * 4 : 0);
  float4_t xv0[XJ], gv[XJ];
  u32x4 idv[SHUF ? XJ : 1];  // act-order: indices of this lane's activations, elements kbase + 4 lane + 256 j + 0..3
  float4_t xa[SHUF ? XJ : 1], ga[SHUF ? XJ : 1];  // act-order: this thread's share of the whole vector, natural order
  if constexpr (SHUF) {
    const rsrc_t ri = make_rsrc(shuffle + kbase, xlen * 4);
#pragma unroll
    for (int j = 0; j < XJ; ++j) idv[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, v16 + j * 1024, 0, 0);
    // chunk c = tid + j * threads (four elements each): K / 4 chunks <= XJ per thread (tiles <= nw * TPW)
    const int nthr = (int)blockDim.x;
    const rsrc_t rxa = make_rsrc(x, WOQ_SKIP(7) ? 0 : K * (xdt == 0 ? 4 : 2));
    const rsrc_t rga = make_rsrc(norm ? (const void*)norm_w : x, norm ? K * 4 : 0);
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int c = tid + j * nthr;
      if (xdt == 0) {
        xa[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rxa, c * 16, 0, 0));
      } else {
        const uint2 r = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rxa, c * 8, 0, 0));
        const uint16_t hb[4] = {(uint16_t)r.x, (uint16_t)(r.x >> 16), (uint16_t)r.y, (uint16_t)(r.y >> 16)};
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float fb = bf16_bits_to_f32(hb[i]), fh = f16_bits_to_f32(hb[i]);
          f[i] = xdt == 2 ? fb : fh;
        }
        xa[j] = (float4_t){f[0], f[1], f[2], f[3]};
      }
      ga[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rga, c * 16, 0, 0));
    }
  } else {
    load_row(0, xv0);
#pragma unroll
    for (int j = 0; j < XJ; ++j)
      gv[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, v16 + j * 1024, 0, 0));
  }
  WOQ_STAMP(1);

  // ---- 1. scales / zero points, then the first PF weight tiles ----
  u32x4 w[CB][TPW];
  typename RawSc<SMODE, S32>::type rsc[CB][TPW];
  uint32_t rzp[CB][TPW];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int tn = (int)blockIdx.x * CB + cb;
    if constexpr (SMODE == 0) {
      const rsrc_t rs = make_rsrc((const char*)scales + (size_t)tn * n_groups * 16 * ESZ, n_groups * 16 * ESZ);
      const rsrc_t rz = make_rsrc(ASYM ? zp + (size_t)tn * n_groups * 16 : (const uint8_t*)scales,
                                  ASYM ? n_groups * 16 : 0);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int grp = min((kt0 + t) >> tpg_shift, n_groups - 1);
        if constexpr (S32)
          rsc[cb][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i16 * 4, grp * 64, 0));
        else
          rsc[cb][t] = __builtin_amdgcn_raw_buffer_load_b16(rs, i16 * 2, grp * 32, 0);
        if constexpr (ASYM) rzp[cb][t] = __builtin_amdgcn_raw_buffer_load_b8(rz, i16, grp * 16, 0);
      }
    } else {
      const rsrc_t rs = make_rsrc((const char*)scales + (size_t)tn * tiles_k * 64 * ESZ, tiles_k * 64 * ESZ);
      const rsrc_t rz = make_rsrc(ASYM ? zp + (size_t)tn * tiles_k * 64 : (const uint8_t*)scales,
                                  ASYM ? tiles_k * 64 : 0);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if constexpr (S32)
          rsc[cb][t] = __builtin_bit_cast(
              float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, i16 * 16 + t * 256, kt0 * 256, 0));
        else
          rsc[cb][t] =
              __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, i16 * 8 + t * 128, kt0 * 128, 0));
        if constexpr (ASYM) rzp[cb][t] = __builtin_amdgcn_raw_buffer_load_b32(rz, i16 * 4 + t * 64, kt0 * 64, 0);
      }
    }
  }
  WOQ_STAMP(2);
  constexpr int PF = WOQ_PF;
  rsrc_t rq[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    rq[cb] = make_rsrc(q + (size_t)((int)blockIdx.x * CB + cb) * tiles_k * 64, min(kt0 + cnt, tiles_k) * 1024);
  auto issue_w = [&](int i) {  // i-th tile load in consumption order (t major, cb minor)
    const int t = i / CB, cb = i % CB;
    w[cb][t] = __builtin_amdgcn_raw_buffer_load_b128(rq[cb], v16 + t * 1024, kt0 * 1024, AUX_NT);
  };
#pragma unroll
  for (int i = 0; i < PF && i < CB * TPW; ++i) issue_w(i);
  WOQ_STAMP(3);

  float ss_coop = 0.f;  // act-order: this wave's share of sum x^2 over the whole vector (any partition sums to the same)
  if constexpr (SHUF) {
    float* xs = (float*)(smem_raw + ((tile_lds_bytes(1, nw, TPW, CB) + 15) & ~(size_t)15));  // [K] fp32: x * norm weight
    const int nthr = (int)blockDim.x;
    const float one = norm ? 0.f : 1.f;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int c = tid + j * nthr;
      ss_coop = fmaf(xa[j].x, xa[j].x, fmaf(xa[j].y, xa[j].y, fmaf(xa[j].z, xa[j].z, fmaf(xa[j].w, xa[j].w, ss_coop))));
      if (c * 4 < K) *(float4_t*)(xs + c * 4) = xa[j] * (ga[j] + one);  // no norm: ga reads as 0 (empty descriptor)
    }
    ss_coop = wave_sum_dpp(ss_coop);
    __syncthreads();
    // this lane's 4 XJ elements of the wave's slice, by index; elements past the slice are zero
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      float f[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool live = lane * 4 + j * 256 + i < xlen;
        // the blob's shuffle vector is foreign data (a checkpoint's g_idx): an index outside [0, K) must not read LDS
        // out of range — clamp it (the result is then wrong for that row only, never an out-of-bounds access)
        const int id = live ? min(max((int)idv[j][i], 0), K - 1) : 0;
        const float v = xs[id];
        f[i] = live ? v : 0.f;
      }
      xv0[j] = (float4_t){f[0], f[1], f[2], f[3]};
      gv[j] = (float4_t){1.f, 1.f, 1.f, 1.f} * (norm ? 1.f : 0.f);  // stage_row multiplies by gv + addone = 1
    }
  }
  // ---- 2. stage this wave's K slice of the activation rows as three int8 limb rows in its LDS strip ----
  ((uint32_t*)zero_blk)[lane] = 0u;
  ((uint32_t*)ones_blk)[lane] = 0x01010101u;
  float my_ss = 0.f;    // lane m (< M) keeps the sum of squares of row m over this slice
  float my_unsc = 0.f;  // lane m keeps 2^(e - 21 - 4): undoes the fixed-point scale and the 16*q weights
  const float addone = norm ? 0.f : 1.f;
  auto stage_row = [&](int m, float4_t (&xv)[XJ]) {
    float ss = 0.f, amax = 0.f;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      ss = fmaf(xv[j].x, xv[j].x, fmaf(xv[j].y, xv[j].y, fmaf(xv[j].z, xv[j].z, fmaf(xv[j].w, xv[j].w, ss))));
      xv[j] = xv[j] * (gv[j] + addone);  // no norm: gv reads as 0 (empty descriptor) and addone = 1
      amax = fmaxf(fmaxf(amax, fabsf(xv[j].x)), fmaxf(fabsf(xv[j].y), fmaxf(fabsf(xv[j].z), fabsf(xv[j].w))));
    }
    amax = wave_max_dpp(amax);
    ss = wave_sum_dpp(ss);
    // amax < 2^e  =>  |x| * 2^(21-e) < 2^21: the sum with 1.5 * 2^23 stays well inside [2^23, 2^24), where
    // ulp = 1, so the fp32 mantissa of the sum IS round(x * 2^(21-e)) + 2^22 (offset binary, 22 bits + sign; the
    // spare bit keeps a round-up at the top of the range from carrying into the exponent)
    int e = 0;
    if (amax > 0.f && amax < INFINITY) e = max(-100, min(100, __builtin_amdgcn_frexp_expf(amax)));
    const float sfix = ldexpf(1.f, 21 - e);
    if (lane == m) {
      my_ss = ss;
      my_unsc = ldexpf(1.f, e - 25);
    }
    unsigned char* r0 = strip + (size_t)(3 * m) * RB + lane * 4;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const uint32_t a = __float_as_uint(fmaf(xv[j].x, sfix, 12582912.f));
      const uint32_t b = __float_as_uint(fmaf(xv[j].y, sfix, 12582912.f));
      const uint32_t c = __float_as_uint(fmaf(xv[j].z, sfix, 12582912.f));
      const uint32_t d = __float_as_uint(fmaf(xv[j].w, sfix, 12582912.f));
      // gather byte n of (a, b, c, d) into one word: v_perm_b32(hi, lo, sel) picks bytes of {hi:lo}, 0x0c = zero
      const uint32_t ab0 = __builtin_amdgcn_perm(b, a, 0x0c0c0400u), cd0 = __builtin_amdgcn_perm(d, c, 0x04000c0cu);
      const uint32_t ab1 = __builtin_amdgcn_perm(b, a, 0x0c0c0501u), cd1 = __builtin_amdgcn_perm(d, c, 0x05010c0cu);
      const uint32_t ab2 = __builtin_amdgcn_perm(b, a, 0x0c0c0602u), cd2 = __builtin_amdgcn_perm(d, c, 0x06020c0cu);
      *(uint32_t*)(r0 + j * 256) = (ab0 | cd0) ^ 0x80808080u;       // limb 0: b0 - 128
      *(uint32_t*)(r0 + RB + j * 256) = (ab1 | cd1) ^ 0x80808080u;  // limb 1: b1 - 128
      *(uint32_t*)(r0 + 2 * RB + j * 256) = ab2 | cd2;              // limb 2: b2 in [0, 127]
    }
  };
#pragma unroll 1
  for (int rs = 0; rs < nrs; ++rs) {  // row sets: rows 4 rs .. 4 rs + Mrs - 1 against the weight tiles in registers
  const int Mrs = M1 ? 1 : min(TSETM, M - rs * TSETM);
  // (the strip is wave-private and LDS executes one wave's accesses in order: the previous set's operand reads are
  // behind us when this set's limbs are written; the barrier only pins the compiler)
  __builtin_amdgcn_wave_barrier();
  my_ss = 0.f, my_unsc = 0.f;
  if (rs == 0) {
    if (!WOQ_SKIP(5)) stage_row(0, xv0);
    if constexpr (SHUF) {
      if (lane == 0) my_ss = ss_coop;  // the gathered slice holds x * g, not x: the RMSNorm sum comes from the copy pass
    }
  } else {
    float4_t xv[XJ];
    load_row((size_t)(rs * TSETM) * lda, xv);
    stage_row(0, xv);
  }
  if constexpr (!M1) {
    for (int m = 1; m < Mrs; ++m) {  // further rows (small batches): loaded behind the weights
      float4_t xv[XJ];
      load_row((size_t)(rs * TSETM + m) * lda, xv);
      stage_row(m, xv);
    }
  }
  if (lane < TSETM) ssq[((size_t)rs * nw + wid) * TSETM + lane] = my_ss;
  // the strip, zero and ones blocks are wave-private and LDS executes one wave's accesses in order: no barrier
  __builtin_amdgcn_wave_barrier();
  WOQ_STAMP(4);
  if constexpr (WOQ_REST_EARLY) {
#pragma unroll
    for (int i = PF; i < CB * TPW; ++i) issue_w(i);
  }

  // ---- 3. inner products, tiles in arrival order ----
  // A rows: MFMA row r = lane & 15 -> activation row r >> 2, part r & 3 (limb 0..2 | ones). D: lane group kq
  // holds rows 4*kq .. 4*kq+3 = everything of activation row kq, for column lane & 15.
  const int a_m = i16 >> 2, a_part = i16 & 3;
  const bool a_live = a_m < Mrs;
  const unsigned char* a_base = !a_live ? zero_blk + kq * 16
                                        : (a_part == 3 ? ones_blk + kq * 16
                                                       : strip + (size_t)(3 * a_m + a_part) * RB + kq * 16);
  const int a_step_t = (a_live && a_part != 3) ? 128 : 0, a_step_h = (a_live && a_part != 3) ? 64 : 0;
  // group-32 scales: the two 32-k halves of a 64-k MFMA belong to different groups -> issue it twice, the
  // operand of the other half read from the zero block
  const unsigned char* a_lo = kq < 2 ? a_base : zero_blk;
  const unsigned char* a_hi = kq < 2 ? zero_blk : a_base;
  const int st_lo_t = kq < 2 ? a_step_t : 0, st_lo_h = kq < 2 ? a_step_h : 0;
  const int st_hi_t = kq < 2 ? 0 : a_step_t, st_hi_h = kq < 2 ? 0 : a_step_h;
  // group-32, batch 1: MFMA rows 4..7 are free, so they carry the SAME activation row restricted to the second 32-k
  // group of the 64-k block (lanes kq >= 2) while rows 0..3 keep the first group (lanes kq < 2): ONE MFMA per 64-k
  // half leaves group 2h in output rows 0..3 (lane quarter 0) and group 2h+1 in rows 4..7 (lane quarter 1)
  const int g_sel = i16 >> 2;
  const bool g_mine = (g_sel == 0 && kq < 2) || (g_sel == 1 && kq >= 2);
  const unsigned char* a_g32 = !g_mine ? zero_blk + kq * 16
                                       : (a_part == 3 ? ones_blk + kq * 16 : strip + (size_t)a_part * RB + kq * 16);
  const int g32_step_t = (g_mine && a_part != 3) ? 128 : 0, g32_step_h = (g_mine && a_part != 3) ? 64 : 0;
  const float unsc = __shfl(my_unsc, M1 && SMODE == 1 ? 0 : kq, 64);
  const i32x4 izero = {0, 0, 0, 0};
  const i32x4 b_ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  float tot[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) tot[cb] = 0.f;
  i32x4 prev_d = izero;  // software pipeline of the per-tile recombination (group-128 path)
  float prev_sc = 0.f, prev_zc = 0.f;
  int prev_cb = 0;
  bool have_prev = false;
  if (WOQ_SKIP(6)) {
    uint32_t acc_ = 0;
#pragma unroll
    for (int i = PF; i < CB * TPW; ++i) issue_w(i);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc_ |= w[cb][t].x | w[cb][t].y | w[cb][t].z | w[cb][t].w;
    tot[0] = acc_ == 0x1234567u ? 1.f : 0.f;
  } else
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if constexpr (NDIG > 0 && !M1) {
      // the digit planes of a tile do not depend on the row set: left alone, the compiler computes all of them once in
      // front of the row-set loop (up to 192 registers) and spills. Opaque per iteration: they are rebuilt per row set.
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) asm volatile("" : "+v"(w[cb][t]));
    }
    if constexpr (SMODE == 0) {
      const i32x4 a0 = *(const i32x4*)(a_base + t * a_step_t);
      const i32x4 a1 = *(const i32x4*)(a_base + t * a_step_t + a_step_h);
      float sx = 0.f;  // sum_k x~_k of this tile (asym only), in fixed-point units
      if constexpr (ASYM) {
        i32x4 ds = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b_ones, izero, 0, 0, 0);
        ds = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b_ones, ds, 0, 0, 0);
        sx = limb_combine(ds);
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        if (!WOQ_REST_EARLY && t * CB + cb + PF < CB * TPW) issue_w(t * CB + cb + PF);
        const u32x4 wv = w[cb][t];
        if constexpr (NDIG > 0) {  // table weights: one MFMA pair per digit plane, recombined most significant first
          i32x4 b0[NB], b1[NB];
          lut_b<NDIG>(lut, wv.x, wv.y, b0);
          lut_b<NDIG>(lut, wv.z, wv.w, b1);
          float f = 0.f;
#pragma unroll
          for (int j = NB - 1; j >= 0; --j) {
            i32x4 d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0[j], izero, 0, 0, 0);
            d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1[j], d, 0, 0, 0);
            f = fmaf(f, 256.f, limb_combine(d));
          }
          float scv;
          if constexpr (S32)
            scv = rsc[cb][t];
          else
            scv = tscale16(rsc[cb][t], bf);
          tot[cb] = fmaf(scv, f, tot[cb]);
          continue;
        }
        const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u),
                          (int)((wv.y << 4) & 0xf0f0f0f0u), (int)(wv.y & 0xf0f0f0f0u)};
        const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u),
                          (int)((wv.w << 4) & 0xf0f0f0f0u), (int)(wv.w & 0xf0f0f0f0u)};
        i32x4 d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, izero, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, d, 0, 0, 0);
        // the recombination of the PREVIOUS tile runs under this tile's MFMA latency
        if (have_prev) tot[prev_cb] = fmaf(prev_sc, limb_combine(prev_d) + prev_zc, tot[prev_cb]);
        prev_d = d;
        prev_cb = cb;
        have_prev = true;
        // b_ones carries 1 where the weights carry 16*q: the zero point enters as 16 * zp
        if constexpr (ASYM) prev_zc = -16.f * (float)((int)(rzp[cb][t] & 0xff) - 8) * sx;
        if constexpr (S32)
          prev_sc = rsc[cb][t];
        else
          prev_sc = tscale16(rsc[cb][t], bf);
      }
    } else if constexpr (M1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const i32x4 av = *(const i32x4*)(a_g32 + t * g32_step_t + h * g32_step_h);
        float sx = 0.f;
        if constexpr (ASYM) sx = limb_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(av, b_ones, izero, 0, 0, 0));
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          if (h == 0 && !WOQ_REST_EARLY && t * CB + cb + PF < CB * TPW) issue_w(t * CB + cb + PF);
          const u32x4 wv = w[cb][t];
          const uint32_t w0 = h == 0 ? wv.x : wv.z, w1 = h == 0 ? wv.y : wv.w;
          float f = half_dot(av, w0, w1);
          // this lane quarter's group: s = 2h + (kq & 1) (quarters 2, 3 hold zeros: their A rows are dead)
          if constexpr (ASYM) {
            const uint32_t zb = (rzp[cb][t] >> (16 * h)) & 0xffffu;
            const int uz = (int)((kq & 1) ? (zb >> 8) : (zb & 0xffu));
            f = fmaf(-16.f * (float)(uz - 8), sx, f);
          }
          float scv;
          if constexpr (S32) {
            const float s_lo = h == 0 ? rsc[cb][t].x : rsc[cb][t].z, s_hi = h == 0 ? rsc[cb][t].y : rsc[cb][t].w;
            scv = (kq & 1) ? s_hi : s_lo;
          } else {
            const uint32_t r = h == 0 ? rsc[cb][t].x : rsc[cb][t].y;
            scv = tscale16((kq & 1) ? (r >> 16) : (r & 0xffffu), bf);
          }
          tot[cb] = fmaf(scv, f, tot[cb]);
        }
      }
    } else {
      i32x4 al[2], ah[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        al[h] = *(const i32x4*)(a_lo + t * st_lo_t + h * st_lo_h);
        ah[h] = *(const i32x4*)(a_hi + t * st_hi_t + h * st_hi_h);
      }
      float sx[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (ASYM) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          sx[2 * h] = limb_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(al[h], b_ones, izero, 0, 0, 0));
          sx[2 * h + 1] = limb_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(ah[h], b_ones, izero, 0, 0, 0));
        }
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        if (!WOQ_REST_EARLY && t * CB + cb + PF < CB * TPW) issue_w(t * CB + cb + PF);
        const u32x4 wv = w[cb][t];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t w0 = h == 0 ? wv.x : wv.z, w1 = h == 0 ? wv.y : wv.w;
          i32x4 b[NB];
          if constexpr (NDIG > 0)
            lut_b<NDIG>(lut, w0, w1, b);
          else
            int4_b(w0, w1, b[0]);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const int s = 2 * h + g2;  // 32-k block of the tile
            const i32x4& ag = g2 == 0 ? al[h] : ah[h];
            float f = limb_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(ag, b[NB - 1], izero, 0, 0, 0));
#pragma unroll
            for (int j = NB - 2; j >= 0; --j)
              f = fmaf(f, 256.f, limb_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(ag, b[j], izero, 0, 0, 0)));
            if constexpr (ASYM) f = fmaf(-16.f * (float)((int)((rzp[cb][t] >> (8 * s)) & 0xff) - 8), sx[s], f);
            float scv;
            if constexpr (S32) {
              scv = rsc[cb][t][s];
            } else {
              const uint32_t r = s < 2 ? rsc[cb][t].x : rsc[cb][t].y;
              scv = tscale16((s & 1) ? (r >> 16) : (r & 0xffff), bf);
            }
            tot[cb] = fmaf(scv, f, tot[cb]);
          }
        }
      }
    }
    if (t == 0) WOQ_STAMP(5);
  }
  if (have_prev) tot[prev_cb] = fmaf(prev_sc, limb_combine(prev_d) + prev_zc, tot[prev_cb]);
  if constexpr (M1 && SMODE == 1) {  // lane quarter 1 holds the odd 32-k groups' sums: fold them into quarter 0
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) tot[cb] += __shfl_xor(tot[cb], 16, 64);
  }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    // slab layout [row set][wave][cb][activation row of the set = lane >> 4][16 columns]
    if (!M1 || lane < 16)
      slab[(((size_t)rs * nw + wid) * CB + cb) * 64 + lane] = tot[cb] * (NDIG > 0 ? unsc * lut.wmul : unsc);
  }
  WOQ_STAMP(6);
  if (WOQ_SKIP(8)) {
    if (lane < 16 && wid == 0) ((float*)out)[(int)blockIdx.x * 16 + lane] = tot[0];
    return;
  }
  }  // row sets
  __syncthreads();
  WOQ_STAMP(7);

  // ---- 4. finish: sum over waves, RMSNorm factor, bias, SiLU*mul, residual, store ----
  const int ncb = silu ? 1 : CB;
  for (int idx = tid; idx < ncb * M * 16; idx += (int)blockDim.x) {
    const int e_i = idx & 15;
    const int e_m = M1 ? 0 : (idx >> 4) % M;
    const int e_cb = M1 ? (idx >> 4) : idx / (16 * M);
    const int e_rs = e_m / TSETM, e_m4 = e_m % TSETM;  // row set, row inside it
    const int slot = e_m4 * 16 + e_i;
    float v = 0.f, up = 0.f, sq = 0.f;
#pragma unroll 4
    for (int w2 = 0; w2 < nw; ++w2) {
      v += slab[(((size_t)e_rs * nw + w2) * CB + e_cb) * 64 + slot];
      if constexpr (CB == 2) up += slab[(((size_t)e_rs * nw + w2) * CB + 1) * 64 + slot];
      sq += ssq[((size_t)e_rs * nw + w2) * TSETM + e_m4];
    }
    const float inv = norm ? 1.0f / sqrtf(sq / (float)K + eps) : 1.f;  // HF LlamaRMSNorm
    v *= inv;
    int n;
    if (silu) {
      n = (int)blockIdx.x * 16 + e_i;
      up *= inv;
      if (bias) {
        v += bias[min(((int)blockIdx.x * 2) * 16 + e_i, N - 1)];
        up += bias[min(((int)blockIdx.x * 2 + 1) * 16 + e_i, N - 1)];
      }
      v = v / (1.0f + __expf(-v)) * up;
    } else {
      n = ((int)blockIdx.x * CB + e_cb) * 16 + e_i;
      if (bias) v += bias[min(n, N - 1)];
    }
    if (n < (silu ? (N >> 1) : N)) {
      if constexpr (M1)
        v += e_res;
      else if (residual)
        v += residual[(size_t)e_m * ld_res + n];
      store_f32(out, (size_t)e_m * ldo + n, out_dtype, v);
    }
  }
  WOQ_STAMP(8);
}

struct TileLaunch {
  const void* q;
  const void* scales;
  const void* zp;
  const void* x;
  const float* norm_w;
  int tiles_k, K, N, n_groups, tpg_shift, M, lda, ldo, ld_res, out_dtype, flags;
  void* out;
  const float* bias;
  const float* residual;
  float eps;
  int nw, grid;
  int kt_begin, kt_count;  // K tiles [kt_begin, kt_begin + kt_count) of the blob covered by this launch
  LutArgs lut;             // table weight types (ndig > 0)
  int ndig;
  const int32_t* shuffle;  // GPTQ act-order: activation index of every weight row (batch 1, int4), or null
};

template <int TPW, int CB, int SMODE, bool ASYM, bool S32, bool M1, int NDIG, bool SHUF = false>
static int launch_tile_t(const TileLaunch& a, hipStream_t st) {
  // act-order form: + the whole activation vector as fp32 behind the regular regions
  const size_t lds = SHUF ? ((tile_lds_bytes(a.M, a.nw, TPW, CB) + 15) & ~(size_t)15) + (size_t)a.K * 4
                          : tile_lds_bytes(a.M, a.nw, TPW, CB);
  if (lds > 160 * 1024) return woq::fail("QBits: activation rows do not fit LDS");
  auto kern = gemv_tile_kernel<TPW, CB, SMODE, ASYM, S32, M1, NDIG, SHUF>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  const int base = a.kt_count / a.nw, rem = a.kt_count % a.nw;
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(a.nw * 64), lds, st, (const u32x4*)a.q, a.scales, a.x, a.norm_w,
                     a.tiles_k, a.K, base, rem, a.n_groups, a.tpg_shift, (const uint8_t*)a.zp, a.out, a.bias,
                     a.residual, a.eps, a.N, a.M, a.lda, a.ldo, a.ld_res, a.out_dtype, a.flags, a.kt_begin, a.lut,
                     a.shuffle);
  return 0;
}

template <int TPW, int CB>
static int launch_tile_sm(const TileLaunch& a, int smode, bool asym, bool s32, hipStream_t st) {
  if (a.shuffle != nullptr) {  // act-order blobs: batch 1, int4 (gemv_tile_max_rows admits nothing else)
    if (a.M != 1 || a.ndig != 0) return woq::fail("QBits: the act-order tile GEMV takes one int4 row");
#define WOQ_TILE_SHUF(SM, AS, S3) \
  if (smode == SM && asym == AS && s32 == S3) return launch_tile_t<TPW, CB, SM, AS, S3, true, 0, true>(a, st);
    WOQ_TILE_SHUF(0, false, false)
    WOQ_TILE_SHUF(0, false, true)
    WOQ_TILE_SHUF(0, true, false)
    WOQ_TILE_SHUF(0, true, true)
    WOQ_TILE_SHUF(1, false, false)
    WOQ_TILE_SHUF(1, false, true)
    WOQ_TILE_SHUF(1, true, false)
    WOQ_TILE_SHUF(1, true, true)
#undef WOQ_TILE_SHUF
    return woq::fail("QBits: bad tile GEMV configuration");
  }
#define WOQ_TILE_CASE(SM, AS, S3, ND)                                                   \
  if (smode == SM && asym == AS && s32 == S3 && a.ndig == ND)                           \
    return a.M == 1 ? launch_tile_t<TPW, CB, SM, AS, S3, true, ND>(a, st)               \
                    : launch_tile_t<TPW, CB, SM, AS, S3, false, ND>(a, st);
  WOQ_TILE_CASE(0, false, false, 0)
  WOQ_TILE_CASE(0, false, true, 0)
  WOQ_TILE_CASE(0, true, false, 0)
  WOQ_TILE_CASE(0, true, true, 0)
  WOQ_TILE_CASE(1, false, false, 0)
  WOQ_TILE_CASE(1, false, true, 0)
  WOQ_TILE_CASE(1, true, false, 0)
  WOQ_TILE_CASE(1, true, true, 0)
  // 4-bit table types: symmetric only; one digit plane (fp4_e2m1), two (bitsandbytes fp4; nf4 at reduced-precision
  // compute) or three (nf4 at compute fp32)
  WOQ_TILE_CASE(0, false, false, 1)
  WOQ_TILE_CASE(0, false, true, 1)
  WOQ_TILE_CASE(1, false, false, 1)
  WOQ_TILE_CASE(1, false, true, 1)
  WOQ_TILE_CASE(0, false, false, 2)
  WOQ_TILE_CASE(0, false, true, 2)
  WOQ_TILE_CASE(1, false, false, 2)
  WOQ_TILE_CASE(1, false, true, 2)
  WOQ_TILE_CASE(0, false, false, 3)
  WOQ_TILE_CASE(0, false, true, 3)
  WOQ_TILE_CASE(1, false, false, 3)
  WOQ_TILE_CASE(1, false, true, 3)
#undef WOQ_TILE_CASE
  return woq::fail("QBits: bad tile GEMV configuration");
}

// Digit planes of a table weight type (woq_gemv_common.h): v = round(table[c] * S) = d0 + 2^8 d1 + 2^16 d2, balanced
// (d0, d1 in [-128, 127]). S: fp4_e2m1 2 (integers up to 12: one digit); the bitsandbytes fp4 table 192 (integers up
// to 192: two digits, exact); nf4 2^22 (three digits, |d2| <= 64) for compute fp32, 127 * 256 (two digits, |d1| <= 127)
// for the reduced-precision compute modes.
int lut_args_for(uint32_t weight_type, uint32_t compute_type, LutArgs& L) {
  memset(&L, 0, sizeof(L));
  L.wmul = 1.f;
  if (!is_table_type(weight_type)) return 0;
  static const float nf4[16] = WOQ_LUT_NF4;
  static const float e2m1[16] = WOQ_LUT_FP4_E2M1;
  static const float bnb[16] = WOQ_LUT_FP4_BNB;
  const float* tab = weight_type == WOQ_W_NF4 ? nf4 : (weight_type == WOQ_W_FP4_E2M1 ? e2m1 : bnb);
  const int ndig = weight_type == WOQ_W_FP4_E2M1 ? 1 : (weight_type == WOQ_W_NF4 && compute_type == WOQ_C_FP32 ? 3 : 2);
  const double S = weight_type == WOQ_W_FP4_E2M1 ? 2.0 : (weight_type == WOQ_W_FP4_E2M1_BNB ? 192.0 : (ndig == 3 ? 4194304.0 : 32512.0));
  for (int c = 0; c < 16; ++c) {
    long v = lrint((double)tab[c] * S);
    for (int j = 0; j < 3; ++j) {
      const int dj = (int)(int8_t)(uint8_t)(v & 0xff);  // low byte, sign-extended
      L.d[j][c >> 2] |= (uint32_t)(uint8_t)dj << (8 * (c & 3));
      v = (v - dj) >> 8;
    }
  }
  L.wmul = (float)(16.0 / S);
  return ndig;
}

// K ranges one launch cannot hold are split into equal chunks run as chained launches (chunk i + 1 adds onto chunk
// i's fp32 output through the residual input): linear epilogues only. Returns the number of chunks, 0 = not covered.
int gemv_tile_k_chunks(int tiles_k, int cb, int smode, bool chainable) {
  int nw, tpw;
  if (gemv_tile_geometry(tiles_k, cb, smode, nw, tpw)) return 1;
  if (!chainable) return 0;
  for (int s = 2; s <= 8; ++s)
    if (gemv_tile_geometry((tiles_k + s - 1) / s, cb, smode, nw, tpw)) return s;
  return 0;
}

// geometry pick: nw waves x tpw tiles cover tiles_k (8 tiles = 8 KiB per wave per column tile; 4 for short K so
// that a workgroup still has a few waves). Returns false when this kernel does not take the shape (K > 16384).
bool gemv_tile_geometry(int tiles_k, int cb, int smode, int& nw, int& tpw) {
  static const int force4 = [] {  // WOQ_TILE_TPW4=<max tiles_k>: 4 tiles per wave up to that K (timing experiments)
    const char* s = getenv("WOQ_TILE_TPW4");
    return s ? atoi(s) : 0;
  }();
  tpw = (tiles_k > 16 && !(cb == 2 && smode == 1) && tiles_k > force4) ? 8 : 4;  // per-32 scales x 2 column tiles: register budget
  nw = (tiles_k + tpw - 1) / tpw;
  return nw <= (cb * tpw > 8 ? 8 : (smode == 1 ? 12 : 16));  // the kernel's __launch_bounds__
}

// largest M the tile kernel takes for this call (LDS budget), 0 if it is not covered: the kernel wants unshuffled
// activation rows, fp32 and 16-B aligned (what the decode engine feeds it and what the reference's qbits boundary
// always holds, modules.py:152-154) or fp16 / bf16 and 8-B aligned; anything else goes to the generic kernel in
// woq_gemv.hip.
int gemv_tile_max_rows(const void* act, int act_dtype, int lda, const woq_blob_header& h, const float* norm_w,
                       int epi, int out_dtype) {
  static const bool table_generic = getenv("WOQ_TABLE_GENERIC") != nullptr;  // A/B switch: round 3's fp32 VALU kernel
  const bool table = is_table_type(h.weight_type) && h.off_zp == 0 && !table_generic;
  // act-order (g_idx) blobs: one int4 row per launch through the gather form (round 5); more rows keep the generic kernel
  static const bool shuf_generic = getenv("WOQ_SHUFFLE_GENERIC") != nullptr;  // A/B switch: the fp32 VALU kernel
  if (h.off_shuffle != 0 && (h.weight_type != WOQ_W_INT4_CLIP || shuf_generic)) return 0;
  if ((h.weight_type != WOQ_W_INT4_CLIP && !table) || (h.K & 3) != 0 || (lda & 3) != 0 ||
      (((uintptr_t)act) & (act_dtype == WOQ_F32 ? 15 : 7)) != 0 || (((uintptr_t)norm_w) & 15) != 0)
    return 0;
  const int tiles_k = h.Kpad / WOQ_TILE_K;
  const int cb = epi == 1 ? 2 : 1;
  int nw, tpw;
  const int chunks = gemv_tile_k_chunks(tiles_k, cb, (int)h.scale_mode, epi == 0 && !norm_w && out_dtype == WOQ_F32);
  if (chunks == 0 || !gemv_tile_geometry((tiles_k + chunks - 1) / chunks, cb, (int)h.scale_mode, nw, tpw)) return 0;
  if (h.off_shuffle != 0 && chunks != 1) return 0;  // the act-order form copies the whole vector per launch: one K range
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return 0;  // tiles per group must be a power of two
  }
  static const int cap = [] {  // WOQ_TILE_MAXM=4: one row set per launch (the round-2 behaviour; same-box A/B runs)
    const char* s = getenv("WOQ_TILE_MAXM");
    const int v = s ? atoi(s) : TMAXM;
    return v >= 1 && v <= TMAXM ? v : TMAXM;
  }();
  int m = h.off_shuffle != 0 ? 1 : cap;
  const size_t extra = h.off_shuffle != 0 ? (size_t)h.K * 4 + 16 : 0;  // the act-order form's copy of the vector
  while (m > 0 && tile_lds_bytes(m, nw, tpw, cb) + extra > 150 * 1024) --m;
  return m;
}

// rows 0..M-1 (M <= gemv_tile_max_rows). x: [M, lda]; out: [M, ldo]; residual: [M, ld_res] or null.
int launch_gemv_tile(const void* act, int act_dtype, int lda, int M, const void* blob, const woq_blob_header& h,
                     const float* bias, void* out, int out_dtype, int ldo, const float* norm_w, float eps,
                     const float* residual, int ld_res, int epi, hipStream_t st) {
  TileLaunch a;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = b + h.off_q;
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.K = h.K;
  a.N = h.N;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
  a.n_groups = h.n_groups;
  a.tpg_shift = 0;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    int tpg = h.group / WOQ_TILE_K;
    while (tpg > 1) {
      tpg >>= 1;
      ++a.tpg_shift;
    }
  }
  a.x = act;
  a.lda = lda;
  a.M = M;
  a.out = out;
  a.out_dtype = out_dtype;
  a.ldo = ldo;
  a.ld_res = ld_res;
  a.bias = bias;
  a.norm_w = norm_w;
  a.eps = eps;
  a.residual = residual;
  a.flags = (h.scale_type == WOQ_BF16 ? 1 : 0) | (epi == 1 ? 2 : 0) |
            (act_dtype == WOQ_F16 ? 4 : (act_dtype == WOQ_BF16 ? 8 : 0));
  a.ndig = lut_args_for(h.weight_type, h.compute_type, a.lut);
  a.shuffle = h.off_shuffle ? (const int32_t*)(b + h.off_shuffle) : nullptr;
#ifdef WOQ_PROBE
  a.flags |= ::g_probe_flags;
#endif
  const int tiles_n = h.Npad / WOQ_TILE_N;
  const int cb = epi == 1 ? 2 : 1;
  if (epi == 1 && (tiles_n & 1)) return woq::fail("QBits: fused gate/up weight needs an even number of column tiles");
  const int smode = (int)h.scale_mode;
  const bool asym = a.zp != nullptr, s32 = h.scale_type == WOQ_F32;
  const int chunks = gemv_tile_k_chunks(a.tiles_k, cb, smode, epi == 0 && !norm_w && out_dtype == WOQ_F32);
  if (M > TMAXM || chunks == 0)
    return woq::fail("QBits: shape not covered by the tile GEMV");
  a.grid = tiles_n / cb;
  const int per = (a.tiles_k + chunks - 1) / chunks;
  for (int c = 0; c < chunks; ++c) {
    a.kt_begin = c * per;
    a.kt_count = std::min(per, a.tiles_k - a.kt_begin);
    if (a.kt_count <= 0) break;
    int tpw;
    if (!gemv_tile_geometry(a.kt_count, cb, smode, a.nw, tpw)) return woq::fail("QBits: shape not covered by the tile GEMV");
    if (c > 0) {  // add onto the previous chunk's output
      a.bias = nullptr;
      a.residual = (const float*)out;
      a.ld_res = ldo;
    }
    int rc;
    if (cb == 2)
      rc = tpw == 4 ? launch_tile_sm<4, 2>(a, smode, asym, s32, st) : launch_tile_sm<8, 2>(a, smode, asym, s32, st);
    else
      rc = tpw == 4 ? launch_tile_sm<4, 1>(a, smode, asym, s32, st) : launch_tile_sm<8, 1>(a, smode, asym, s32, st);
    if (rc) return rc;
  }
  return 0;
}

}  // namespace woq
