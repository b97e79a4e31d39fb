// woq_gemv_tile.hip — small-M (1..8 rows) int4 GEMV, the per-token hot kernel: "everything in flight at once".
//
// Arithmetic and parity definition (reference): qbits.cpp:113-140 -> bestla_weightonly_dispatcher.cpp:120-189
// (per N-tile x K-block: unpack int4, apply scale / zero point, fp32 accumulate, epilogue
// alpha*acc + beta*bias, bestla_customop.hpp:22-40); definition autograd/functions.py:41-63.
//
// Schedule. A batch-1 projection is 8-45 MB of weights read exactly once; at 8 TB/s that is 1-6 us, so the
// kernel is a latency chain, not a loop, and what it must NOT do is spend instructions: measured on MI355X
// (profiles/r01_gemv_probe_*.txt) a first version with 4 tiles per wave was ISSUE-bound at ~4 B per wave
// instruction = 4.7 TB/s with 3.4 us of fixed cost per launch. Hence:
//  * one workgroup = CB adjacent 16-column tiles (CB = 2 for the fused gate/up SiLU*mul pairs) x all of K;
//    its few, fat waves own contiguous balanced K slices of up to TPW = 8..16 tiles (8-16 KiB) each, so the
//    per-wave fixed cost (addresses, activation staging, reduction) is paid once per 16 KiB;
//  * a wave issues the activation-row loads first (they come back from L2 first: loads return in order), then
//    ALL of its scale / zero-point loads and its CB*TPW 1-KiB tile loads (global_load_dwordx4 nt, straight to
//    VGPRs) before anything waits: weights do not depend on activations, so the whole matrix is in flight a
//    few hundred ns after dispatch and HBM sees one deep queue;
//  * while the weights fly, the wave stages ONLY ITS OWN K slice of the activation rows into a wave-private
//    LDS strip as hi/lo fp16 — no workgroup barrier, no full-vector dependency:
//      - RMSNorm is separable: out = rsqrt(mean(x^2)+eps) * (W . (x*g)); every wave adds its slice's sum of
//        squares to the reduction slab and the factor is applied in the epilogue;
//      - the fp16 range scale 2^-e is chosen per wave from its slice's max and divided out of its partials;
//  * inner product on the matrix pipe (v_mfma_f32_16x16x32_f16): B = the dequantised 16-column x 32-k fragment
//    (a blob tile IS four such fragments: packed-fp16 magic-number dequantisation, 9 VALU per 8 weights),
//    A = activation rows, row 2m = hi = fp16_rtz(x_m 2^-e), row 2m+1 = lo = fp16(x_m 2^-e - hi): both partial
//    products are exact in fp32 and their sum carries x to ~2^-21 relative — fp32-class accuracy for fp32
//    activations. Up to 8 rows ride along for free; tiles are consumed in arrival order (counted vmcnt);
//  * ONE barrier per workgroup: partial sums -> LDS slab -> the first CB*M*16 threads finish
//    (RMSNorm factor, bias, residual, SiLU*mul) and store.
// Kernel arguments are plain scalars (not a struct) so the first 14 dwords are preloaded into SGPRs at dispatch
// (-mllvm -amdgpu-kernarg-preload-count, see the Makefile): no s_load round trip before the first address.
#include "woq_device.h"
#include "woq_launch.h"

// Timeline probe hook (tools/gemv_probe.hip compiles this file with WOQ_PROBE): lane 0 of every wave drops a
// 100 MHz wall-clock stamp per stage. Expands to nothing in the product build.
#ifdef WOQ_PROBE
extern __device__ unsigned long long* g_probe;
#define WOQ_STAMP(k)                                                                     \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (g_probe && lane == 0) {                                                          \
      unsigned long long* slot_ = g_probe + ((size_t)blockIdx.x * 16 + wid) * 32;        \
      slot_[(k) + 10 * probe_pass_] = clock64();                                         \
      if ((k) == 0 && probe_pass_ == 0) slot_[31] = wall_clock64();                      \
    }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  } while (0)
#define WOQ_PROBE_PASSES for (int probe_pass_ = 0; probe_pass_ < 2; ++probe_pass_)
#define WOQ_PROBE_PASS_END __syncthreads();
#else
#define WOQ_STAMP(k)
#define WOQ_PROBE_PASSES
#define WOQ_PROBE_PASS_END
#endif

namespace woq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int TMAXM = 8;  // activation rows per launch (16 MFMA rows / (hi, lo))

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

// 8 packed int4 (one u32, interleaved nibble order) -> 8 fp16 values (u - uz), k-consecutive
__device__ __forceinline__ h8 tdq8(uint32_t w, h2 c1024, h2 c64) {
  const uint32_t w8 = w >> 8;
  const uint32_t o0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w & 0x000f000fu) | 0x64006400u) + c1024);
  const uint32_t o1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w & 0x00f000f0u) | 0x54005400u) + c64);
  const uint32_t o2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w8 & 0x000f000fu) | 0x64006400u) + c1024);
  const uint32_t o3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w8 & 0x00f000f0u) | 0x54005400u) + c64);
  return __builtin_bit_cast(h8, (u32x4){o0, o1, o2, o3});
}

// LDS (dynamic): [nw zero blocks of 256 B][nw strips: 2*M rows x (TPW*128 + 8) halves][slab nw x CB x 128 f32]
//                [sumsq nw x TMAXM f32]
__host__ __device__ constexpr int tile_row_halves(int TPW) { return TPW * 128 + 8; }  // +16 B: rows on distinct banks
__host__ __device__ inline size_t tile_lds_bytes(int M, int nw, int TPW, int CB) {
  return (size_t)nw * 256 + (size_t)nw * 2 * M * tile_row_halves(TPW) * 2 + (size_t)nw * CB * 128 * 4 +
         (size_t)nw * TMAXM * 4;
}

// raw buffer descriptor (gfx950: dword3 = 0x00020000, 32-bit raw data format). Out-of-range reads return 0 and
// touch no memory, so slice / matrix edges need no clamps or masks anywhere below. `p` must be wave-uniform.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
constexpr int AUX_NT = 2;  // non-temporal: streamed-once weights

// flags: bit 0 scales are bf16 (else fp16; ignored for fp32 scales), bit 1 SiLU(gate)*up epilogue (CB == 2)
template <int TPW, int CB, int SMODE, bool ASYM, bool S32, bool M1>
__global__ __launch_bounds__(CB * TPW > 8 ? 512 : 1024) void gemv_tile_kernel(
    const u32x4* __restrict__ q, const void* __restrict__ scales, const float* __restrict__ x,
    const float* __restrict__ norm_w, int tiles_k, int K, int base_tiles, int rem_tiles, int n_groups, int tpg_shift,
    const uint8_t* __restrict__ zp, void* __restrict__ out, const float* __restrict__ bias, const float* residual,
    float eps, int N, int Mrows, int lda, int ldo, int ld_res, int out_dtype, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int RH = tile_row_halves(TPW);
  constexpr int XJ = TPW / 2;  // float4 loads per lane per row covering TPW*128 activations
  constexpr int ESZ = S32 ? 4 : 2;
  static_assert((TPW & 1) == 0, "TPW must be even");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int M = M1 ? 1 : Mrows;
  _Float16* zero_blk = (_Float16*)smem_raw + wid * 128;  // this wave's 256 B of zeros
  _Float16* strips = (_Float16*)smem_raw + nw * 128;
  _Float16* strip = strips + (size_t)wid * 2 * M * RH;        // this wave's [2*M][RH]
  float* slab = (float*)(strips + (size_t)nw * 2 * M * RH);  // [nw][CB][128]
  float* ssq = slab + nw * CB * 128;                          // [nw][TMAXM]

  // this wave's K tiles: balanced contiguous slice [kt0, kt0 + cnt), cnt <= TPW. Everything past the slice end
  // reads as zero through the descriptors' bounds, so tiles t >= cnt contribute exactly 0.
  const int kt0 = wid * base_tiles + min(wid, rem_tiles);
  const int cnt = base_tiles + (wid < rem_tiles ? 1 : 0);
  const int i16 = lane & 15;
  const int kbase = kt0 * 128;
  const int xlen = max(0, min(cnt * 128, K - kbase));  // activations of this slice
  const bool norm = norm_w != nullptr;
  const bool bf = (flags & 1) != 0, silu = (flags & 2) != 0;
  const int v16 = lane * 16;
  WOQ_PROBE_PASSES {  // (probe build only: the body runs twice, the second time with warm instruction cache)
  WOQ_STAMP(0);

  // ---- 0. row 0 of the activations (and the RMSNorm weight) first: loads return in order, so these come back
  //         from L2 long before the weight stream and the staging below overlaps the HBM latency ----
  const rsrc_t rx = make_rsrc(x + kbase, xlen * 4);
  const rsrc_t rg = make_rsrc(norm ? norm_w + kbase : x, norm ? xlen * 4 : 0);
  float4_t xv0[XJ], gv[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    xv0[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, v16 + j * 1024, 0, 0));
    gv[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, v16 + j * 1024, 0, 0));
  }

  WOQ_STAMP(1);
  // ---- 1. scales / zero points, then the whole weight slice of this wave: everything is in flight now ----
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
  // The CU's vector-memory queue holds only ~16 KiB of misses (measured: issue of 8 x 1-KiB loads per wave blocks
  // for ~3.6k cycles): a wave that tries to issue its whole slice first sits in the issue stall while its staging
  // and arithmetic wait behind it. So: PF tile loads now, the rest one per consumed tile (same program order as
  // consumption, so every wait stays a counted vmcnt).
  constexpr int PF = 4;
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

  // ---- 2. stage this wave's K slice of the activation rows: (x * g) * 2^-e -> hi / lo fp16 in its LDS strip ----
  ((uint32_t*)zero_blk)[lane] = 0u;
  float my_ss = 0.f;     // lane m (< M) keeps the sum of squares of row m over this slice
  float my_p2inv = 0.f;  // lane m keeps 2^e of row m
  const float addone = norm ? 0.f : 1.f;
  auto stage_row = [&](int m, float4_t (&xv)[XJ]) {
    float ss = 0.f, amax = 0.f;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      ss = fmaf(xv[j].x, xv[j].x, fmaf(xv[j].y, xv[j].y, fmaf(xv[j].z, xv[j].z, fmaf(xv[j].w, xv[j].w, ss))));
      xv[j] = xv[j] * (gv[j] + addone);  // no norm: gv reads as 0 (empty descriptor) and addone = 1
      amax = fmaxf(fmaxf(amax, fabsf(xv[j].x)), fmaxf(fabsf(xv[j].y), fmaxf(fabsf(xv[j].z), fabsf(xv[j].w))));
    }
    ss = wave_sum(ss);
    amax = wave_max(amax);
    // 2^-e with |x| * 2^-e <= 2^13: exact scaling, comfortably inside fp16 range
    int e = 0;
    if (amax > 0.f && amax < INFINITY) e = max(-40, min(40, (int)ceilf(log2f(amax)) - 13));
    const float p2 = exp2f((float)-e);
    if (lane == m) {
      my_ss = ss;
      my_p2inv = exp2f((float)e);
    }
    _Float16* hi = strip + (size_t)(2 * m) * RH + lane * 4;
    _Float16* lo = hi + RH;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const float4_t v = xv[j] * p2;
      // hi = round-toward-zero fp16 (one packed convert per pair); v - hi is exact in fp32
      const fp16x2 ha = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), hb = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
      const fp16x2 la = __builtin_amdgcn_cvt_pkrtz(v.x - (float)ha.x, v.y - (float)ha.y);
      const fp16x2 lb = __builtin_amdgcn_cvt_pkrtz(v.z - (float)hb.x, v.w - (float)hb.y);
      *(uint2*)(hi + j * 256) = (uint2){__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb)};
      *(uint2*)(lo + j * 256) = (uint2){__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb)};
    }
  };
  stage_row(0, xv0);
  if constexpr (!M1) {
    for (int m = 1; m < M; ++m) {  // further rows (small batches): loaded behind the weights
      const rsrc_t rxm = make_rsrc(x + (size_t)m * lda + kbase, xlen * 4);
      float4_t xv[XJ];
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        xv[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rxm, v16 + j * 1024, 0, 0));
      stage_row(m, xv);
    }
  }
  if (lane < TMAXM) ssq[wid * TMAXM + lane] = my_ss;
  // the strip and the zero block are wave-private and LDS executes one wave's accesses in order: no barrier
  __builtin_amdgcn_wave_barrier();
  WOQ_STAMP(4);

  // ---- 3. inner products, tiles in arrival order ----
  // this lane's D rows: 4*(lane>>4) + {0,1} -> activation row 2*(lane>>4); + {2,3} -> row 2*(lane>>4) + 1
  const int ma = 2 * (lane >> 4), mb = ma + 1;
  const float out_scale_a = __shfl(my_p2inv, ma, 64), out_scale_b = __shfl(my_p2inv, mb, 64);
  // this lane's A row: MFMA row r = lane & 15 -> activation row r >> 1, part (hi | lo) r & 1
  const bool a_live = i16 < 2 * M;
  const _Float16* a_base = (a_live ? strip + (size_t)i16 * RH : zero_blk) + (lane >> 4) * 8;
  const int a_step = a_live ? 128 : 0;
  // nibble masks / magic exponents in VGPRs (opaque to the optimiser) so each extract is ONE v_and_or_b32
  uint32_t m_lo, m_hi, c_lo, c_hi;
  asm volatile("v_mov_b32 %0, 0x000f000f" : "=v"(m_lo));
  asm volatile("v_mov_b32 %0, 0x00f000f0" : "=v"(m_hi));
  asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(c_lo));
  asm volatile("v_mov_b32 %0, 0x54005400" : "=v"(c_hi));
  auto dq8 = [&](uint32_t wv, h2 c1024, h2 c64) -> h8 {
    const uint32_t w8 = wv >> 8;
    const uint32_t o0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (wv & m_lo) | c_lo) + c1024);
    const uint32_t o1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (wv & m_hi) | c_hi) + c64);
    const uint32_t o2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w8 & m_lo) | c_lo) + c1024);
    const uint32_t o3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w8 & m_hi) | c_hi) + c64);
    return __builtin_bit_cast(h8, (u32x4){o0, o1, o2, o3});
  };
  float tot_a[CB], tot_b[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) tot_a[cb] = tot_b[cb] = 0.f;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    h8 afr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) afr[s] = *(const h8*)(a_base + t * a_step + s * 32);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      if (t * CB + cb + PF < CB * TPW) issue_w(t * CB + cb + PF);
      float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float z = 8.f;
        if constexpr (ASYM) {
          if constexpr (SMODE == 0)
            z = (float)(rzp[cb][t] & 0xff);
          else
            z = (float)((rzp[cb][t] >> (8 * s)) & 0xff);
        }
        const _Float16 z1 = (_Float16)(-(1024.f + z)), z2 = (_Float16)(-(64.f + z));
        const h8 bfr = dq8(w[cb][t][s], (h2){z1, z1}, (h2){z2, z2});
        if constexpr (SMODE == 0) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[s], bfr, acc, 0, 0, 0);
        } else {
          const float4_t zero = {0.f, 0.f, 0.f, 0.f};
          const float4_t d = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[s], bfr, zero, 0, 0, 0);
          float scv;
          if constexpr (S32) {
            scv = rsc[cb][t][s];
          } else {
            const uint32_t r = s < 2 ? rsc[cb][t].x : rsc[cb][t].y;
            scv = tscale16((s & 1) ? (r >> 16) : (r & 0xffff), bf);
          }
          tot_a[cb] = fmaf(scv, d.x + d.y, tot_a[cb]);
          if constexpr (!M1) tot_b[cb] = fmaf(scv, d.z + d.w, tot_b[cb]);
        }
      }
      if constexpr (SMODE == 0) {
        float scv;
        if constexpr (S32)
          scv = rsc[cb][t];
        else
          scv = tscale16(rsc[cb][t], bf);
        tot_a[cb] = fmaf(scv, acc.x + acc.y, tot_a[cb]);
        if constexpr (!M1) tot_b[cb] = fmaf(scv, acc.z + acc.w, tot_b[cb]);
      }
    }
    if (t == 0) WOQ_STAMP(5);
  }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    // slab layout [wave][cb][row-pair slot lane>>4][a|b][16 columns]
    float* dst = slab + ((size_t)wid * CB + cb) * 128 + (lane >> 4) * 32 + i16;
    if constexpr (M1) {
      if (lane < 16) dst[0] = tot_a[cb] * out_scale_a;
    } else {
      dst[0] = tot_a[cb] * out_scale_a;
      dst[16] = tot_b[cb] * out_scale_b;
    }
  }
  WOQ_STAMP(6);
  __syncthreads();
  WOQ_STAMP(7);

  // ---- 4. finish: sum over waves, RMSNorm factor, bias, SiLU*mul, residual, store ----
  const int ncb = silu ? 1 : CB;
  for (int idx = tid; idx < ncb * M * 16; idx += (int)blockDim.x) {
    const int e_i = idx & 15;
    const int e_m = M1 ? 0 : (idx >> 4) % M;
    const int e_cb = M1 ? (idx >> 4) : idx / (16 * M);
    const int slot = (e_m >> 1) * 32 + (e_m & 1) * 16 + e_i;
    float v = 0.f, up = 0.f, sq = 0.f;
    for (int w2 = 0; w2 < nw; ++w2) {
      v += slab[((size_t)w2 * CB + e_cb) * 128 + slot];
      if constexpr (CB == 2) up += slab[((size_t)w2 * CB + 1) * 128 + slot];
      sq += ssq[w2 * TMAXM + e_m];
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
      if (residual) v += residual[(size_t)e_m * ld_res + n];
      store_f32(out, (size_t)e_m * ldo + n, out_dtype, v);
    }
  }
  WOQ_STAMP(8);
  WOQ_PROBE_PASS_END
  }
}

struct TileLaunch {
  const void* q;
  const void* scales;
  const void* zp;
  const float* x;
  const float* norm_w;
  int tiles_k, K, N, n_groups, tpg_shift, M, lda, ldo, ld_res, out_dtype, flags;
  void* out;
  const float* bias;
  const float* residual;
  float eps;
  int nw, grid;
};

template <int TPW, int CB, int SMODE, bool ASYM, bool S32, bool M1>
static int launch_tile_t(const TileLaunch& a, hipStream_t st) {
  const size_t lds = tile_lds_bytes(a.M, a.nw, TPW, CB);
  if (lds > 160 * 1024) return woq::fail("QBits: activation rows do not fit LDS");
  auto kern = gemv_tile_kernel<TPW, CB, SMODE, ASYM, S32, M1>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  const int base = a.tiles_k / a.nw, rem = a.tiles_k % a.nw;
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(a.nw * 64), lds, st, (const u32x4*)a.q, a.scales, a.x, a.norm_w,
                     a.tiles_k, a.K, base, rem, a.n_groups, a.tpg_shift, (const uint8_t*)a.zp, a.out, a.bias,
                     a.residual, a.eps, a.N, a.M, a.lda, a.ldo, a.ld_res, a.out_dtype, a.flags);
  return 0;
}

template <int TPW, int CB>
static int launch_tile_sm(const TileLaunch& a, int smode, bool asym, bool s32, hipStream_t st) {
#define WOQ_TILE_CASE(SM, AS, S3)                                                                                 \
  if (smode == SM && asym == AS && s32 == S3)                                                                     \
    return a.M == 1 ? launch_tile_t<TPW, CB, SM, AS, S3, true>(a, st) : launch_tile_t<TPW, CB, SM, AS, S3, false>(a, st);
  WOQ_TILE_CASE(0, false, false)
  WOQ_TILE_CASE(0, false, true)
  WOQ_TILE_CASE(0, true, false)
  WOQ_TILE_CASE(0, true, true)
  WOQ_TILE_CASE(1, false, false)
  WOQ_TILE_CASE(1, false, true)
  WOQ_TILE_CASE(1, true, false)
  WOQ_TILE_CASE(1, true, true)
#undef WOQ_TILE_CASE
  return woq::fail("QBits: bad tile GEMV configuration");
}

// geometry pick: nw waves x tpw tiles cover tiles_k (8 tiles = 8 KiB per wave per column tile; 4 for short K so
// that a workgroup still has a few waves). Returns false when this kernel does not take the shape (K > 16384).
static bool tile_geometry(int tiles_k, int cb, int& nw, int& tpw) {
  tpw = tiles_k > 16 ? 8 : 4;
  nw = (tiles_k + tpw - 1) / tpw;
  return nw <= (cb * tpw > 8 ? 8 : 16);  // the kernel's __launch_bounds__
}

// largest M the tile kernel takes for this call (LDS budget), 0 if it is not covered: the kernel wants fp32,
// 16-B aligned, unshuffled activation rows (what the decode engine feeds it and what the reference's qbits
// boundary always holds, modules.py:152-154); anything else goes to the generic kernel in woq_gemv.hip.
int gemv_tile_max_rows(const void* act, int act_dtype, int lda, const woq_blob_header& h, const float* norm_w,
                       int epi) {
  if (act_dtype != WOQ_F32 || h.off_shuffle != 0 || (h.K & 3) != 0 || (lda & 3) != 0 ||
      (((uintptr_t)act) & 15) != 0 || (((uintptr_t)norm_w) & 15) != 0)
    return 0;
  const int tiles_k = h.Kpad / WOQ_TILE_K;
  const int cb = epi == 1 ? 2 : 1;
  int nw, tpw;
  if (!tile_geometry(tiles_k, cb, nw, tpw)) return 0;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return 0;  // tiles per group must be a power of two
  }
  int m = TMAXM;
  while (m > 0 && tile_lds_bytes(m, nw, tpw, cb) > 150 * 1024) --m;
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
  a.x = (const float*)act;
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
  a.flags = (h.scale_type == WOQ_BF16 ? 1 : 0) | (epi == 1 ? 2 : 0);
  const int tiles_n = h.Npad / WOQ_TILE_N;
  const int cb = epi == 1 ? 2 : 1;
  if (epi == 1 && (tiles_n & 1)) return woq::fail("QBits: fused gate/up weight needs an even number of column tiles");
  int tpw;
  if (act_dtype != WOQ_F32 || !tile_geometry(a.tiles_k, cb, a.nw, tpw))
    return woq::fail("QBits: shape not covered by the tile GEMV");
  a.grid = tiles_n / cb;
  const int smode = (int)h.scale_mode;
  const bool asym = a.zp != nullptr, s32 = h.scale_type == WOQ_F32;
  if (cb == 2) return tpw == 4 ? launch_tile_sm<4, 2>(a, smode, asym, s32, st) : launch_tile_sm<8, 2>(a, smode, asym, s32, st);
  return tpw == 4 ? launch_tile_sm<4, 1>(a, smode, asym, s32, st) : launch_tile_sm<8, 1>(a, smode, asym, s32, st);
}

}  // namespace woq
