// woq_gemv_xqm.h — the batch-1 decode GEMV as FEW, LONG-LIVED workgroups (round 6). REJECTED: parity-green and
// 0.6-1.2 us per launch slower than woq_gemv_xqs.h wherever it walks several strips (profiles/r06b_xqm_*); its host
// side (geometry pick, launchers, the fused qkv + attention variant) is in git history at commit 504a3f3.
//
// Arithmetic and parity definition as woq_gemv_xqs.h (reference qbits.cpp:113-140, autograd/functions.py:41-63): exact
// int8 x int8 -> int32 tile sums on v_mfma_i32_16x16x64_i8 over an XQ activation vector (woq_xq.h), one fp32
// recombination per 16-k block, group scale (and zero point) per block, fp32 across blocks.
//
// Why a second kernel. woq_gemv_xqs.h runs one workgroup per 16-column strip: 768 / 688 workgroups for the qkv and
// gate/up projections of Llama-2-7B, three per CU. Its own knock-outs (profiles/r03c_xq_knockouts_lean.txt) say what
// that costs: with EVERY descriptor emptied — no memory traffic at all — the qkv launch still takes 4.6 us and the
// gate/up launch 6.6 us against 2.5-2.8 us for an empty kernel on the same grid, and the bare skeleton (no traffic, no
// arithmetic) 4.1 us: ~215 set-up instructions per wave for 8 tile visits of ~41 instructions, and a per-workgroup
// dispatch cost that follows the workgroup count (skeleton minus empty: 1.4-1.6 us at 688-768 workgroups, 0.6-0.85 us
// at 256). So this kernel cuts the grid to ONE workgroup per CU (n_wg = 256) that walks its strips b, b + n_wg, ...:
//   * a wave owns one K slice for ALL of the workgroup's strips, so its A operand (the limb blocks of that slice) and
//     the block factors are loaded ONCE, straight into registers — no LDS parking, no per-tile ds_read of A / u;
//   * the weight window rolls ACROSS strips: register slot t is re-requested for strip s + 1 the moment tile t of
//     strip s has been consumed, so the stream never drains at a strip (= old workgroup) boundary;
//   * no workgroup barrier behind the stream: a wave leaves its 16 (x CB) partial sums in the strip's slab row and
//     bumps the strip's LDS counter; whichever wave arrives LAST runs that strip's epilogue at once while the others
//     are already in the next strip (LDS executes a wave's operations in order, so the counter is behind the row);
//   * what the epilogue lanes need from memory (residual, next norm weight, RMSNorm partials) is fetched by wave 0 up
//     front for all strips and parked in LDS.
// Scope: int4 weights (the table types keep woq_gemv_xqs.h), every scale mode / zero-point / scale type of it.
#pragma once
#include "woq_comm_dev.h"
#include "woq_gemv_common.h"
#include "woq_gemv_xqs.h"
#include "woq_xq.h"

// weight tiles a wave requests BEFORE its small (L2-resident) requests; returns are in order, so what sits in front of
// the A operand delays the first MFMA (same switch as WOQ_XQS_PRE)
#ifndef WOQ_XQM_PRE
#define WOQ_XQM_PRE 2
#endif

namespace woq {

constexpr int XQM_OOB = 0x40000000;  // an offset no descriptor below covers: the request returns 0 and moves no bytes

// LDS: [zero block 256][per wave: limb strip TPW x 384 | block factors u | block sums sx | scale slices CB x SCB |
// zero-point slices CB x ZPB][slab SMAX x nw x CB x 16 f32][counters SMAX][epilogue inputs SMAX x 2 x 16 f32][sum of squares]
template <int TPW, int CB, int SMODE, bool ASYM, bool S32>
struct XqmLds {
  static constexpr int ESZ = S32 ? 4 : 2;
  static constexpr int STRIP = ((TPW * 384 + 15) / 16) * 16;
  static constexpr int UTAB = ((TPW * 8 + 63) / 64) * 256;  // one fp32 per block of the slice
  static constexpr int SXTAB = ASYM ? UTAB : 0;
  static constexpr int SCB = (SMODE == 0 ? TPW * 16 : TPW * 64) * ESZ;  // one column tile's scale slice of a wave
  static constexpr int ZPB = ASYM ? (SMODE == 0 ? TPW * 16 : TPW * 64) : 0;
  static constexpr int O_U = STRIP, O_SX = O_U + UTAB, O_SC = O_SX + SXTAB, O_ZP = O_SC + CB * SCB;
  static constexpr int WAVE = ((O_ZP + CB * ZPB + 15) / 16) * 16;
  __host__ __device__ static constexpr size_t slab_off(int nw) { return 256 + (size_t)nw * WAVE; }
  __host__ __device__ static constexpr size_t total(int nw, int smax) {
    return 256 + (size_t)nw * WAVE + (size_t)smax * nw * CB * 64 + (((size_t)smax * 4 + 15) / 16) * 16 + (size_t)smax * 128 + 16;
  }
};

__device__ __forceinline__ unsigned int lds_inc(unsigned int* p) {
  // LDS atomic with return, on the local address space (no vmcnt wait: the weight window stays in flight)
  typedef __attribute__((address_space(3))) unsigned int lds_u32;
  return __hip_atomic_fetch_add((lds_u32*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Every vector-memory request on a wave's common path is inline asm with COUNTED waits: hipcc's own wait insertion
// gives up at the joins behind the per-strip epilogue branch (loads and stores of one wave on one side) and drains
// the whole window with vmcnt(0) at every strip (the first build of this kernel did exactly that). Returns are in
// order, so "wait until at most N requests are outstanding" = "everything older than the N youngest has landed";
// requests the compiler itself issues later (the epilogue of the one wave that runs it) only make a wait stricter.
typedef unsigned int xqm_rsrc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ xqm_rsrc xqm_make_rsrc(const void* p, int bytes) {
  const unsigned long long a = (unsigned long long)p;
  return xqm_rsrc{(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a),
                  (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)(a >> 32) & 0xffffu)),
                  (unsigned int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
}
#define XQM_LD128(dst, voff, rsrc, soff) \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff))
#define XQM_LD128_NT(dst, voff, rsrc, soff) \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff))
#define XQM_LD32(dst, voff, rsrc, soff) \
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff))
#define XQM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) > 63 ? 63 : (n)))  // the counter has 6 bits
#define XQM_TOUCH(x) asm volatile("" : "+v"(x))  // what follows reads x AFTER the wait in front of this statement

template <int TPW, int CB, int SMODE, bool ASYM, bool S32, bool FUSED, int SMAX>
__device__ __forceinline__ void gemv_xqm_body(
    unsigned char* smem_raw, const u32x4* __restrict__ q, const void* __restrict__ scales,
    const uint8_t* __restrict__ xlimbs, const float* __restrict__ xu, int tiles_k, int kt_off, int base_tiles,
    int rem_tiles, int n_groups, int tpg_shift, const uint8_t* __restrict__ zp, const float* __restrict__ xsx,
    float* __restrict__ out, const float* __restrict__ bias, const float* residual, float eps, int N, int K, int flags,
    const float* __restrict__ ssq_in, int n_ssq, const XqPtrs& xo, const float* __restrict__ next_norm_w,
    float* __restrict__ ssq_out, const CommDev* __restrict__ tp, int n_strips, int n_wg, unsigned int fused_tag) {
  typedef XqmLds<TPW, CB, SMODE, ASYM, S32> L;
  constexpr int ESZ = L::ESZ;
  static_assert(SMAX >= 1 && SMAX <= 4, "one pass of four 16-lane rows carries the epilogue inputs");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = uni(tid >> 6);
  const int nw = (int)blockDim.x >> 6;
  const int b = (int)blockIdx.x;
  WOQ_XQS_STAMP(0);
  const int ns = uni((n_strips - b + n_wg - 1) / n_wg);  // strips b, b + n_wg, ... of this workgroup
  const int kt0 = kt_off + wid * base_tiles + min(wid, rem_tiles);
  const int cnt = uni(base_tiles + (wid < rem_tiles ? 1 : 0));
  const int v16 = lane * 16;
  const int i16 = lane & 15, kq = lane >> 4;

  unsigned char* zero_blk = smem_raw;
  unsigned char* wbase = smem_raw + 256 + (size_t)wid * L::WAVE;
  float* slab = (float*)(smem_raw + L::slab_off(nw));                                 // [SMAX][nw][CB][16]
  unsigned int* cnts = (unsigned int*)(slab + (size_t)SMAX * nw * CB * 16);           // [SMAX]
  float* epi = (float*)((unsigned char*)cnts + (((size_t)SMAX * 4 + 15) / 16) * 16);  // [SMAX][2][16]
  float* red = epi + (size_t)SMAX * 32;
  const bool silu = (flags & 2) != 0;
  const int n_out = silu ? (N >> 1) : N;

  // ---- 0. wave 0: what the epilogue lanes need from memory, for every strip of the workgroup (16 lanes per strip) and
  // the RMSNorm partials (256 per 16-byte piece). Requested FIRST — L2-resident, and returns are in order — parked in
  // LDS once the window is out, which is before this wave can bump any strip's counter.
  float4_t ssq_v[4];
  float e_res = 0.f, g_next = 1.f;
  if (wid == 0) {
    if (lane < SMAX) cnts[lane] = 0u;
    const xqm_rsrc rss = xqm_make_rsrc(ssq_in, ssq_in != nullptr ? n_ssq * 4 : 0);
    XQM_LD128(ssq_v[0], v16, rss, 0);
    XQM_LD128(ssq_v[1], v16, rss, 1024);
    XQM_LD128(ssq_v[2], v16, rss, 2048);
    XQM_LD128(ssq_v[3], v16, rss, 3072);
    const xqm_rsrc rres = xqm_make_rsrc(residual, residual != nullptr ? n_out * 4 : 0);
    const xqm_rsrc rgn = xqm_make_rsrc(next_norm_w, next_norm_w != nullptr ? n_out * 4 : 0);
    const int vo = kq < ns ? ((b + kq * n_wg) * 16 + i16) * 4 : XQM_OOB;  // past n_out: zeros through the descriptor
    XQM_LD32(e_res, vo, rres, 0);
    XQM_LD32(g_next, vo, rgn, 0);
  }

  // ---- 1. the weight window of strip 0: nothing from HBM in front of its first PRE tiles ----
  const xqm_rsrc rq = xqm_make_rsrc(q, n_strips * CB * tiles_k * 1024);
  // byte offset of this wave's tile 0 of column tile (strip s, cb 0); past the last strip: nowhere
  auto strip_off = [&](int s) -> int { return s < ns ? uni(((b + s * n_wg) * CB * tiles_k + kt0) * 1024) : XQM_OOB; };
  const int cb_step = tiles_k * 1024;
  u32x4 w[CB][TPW];
  constexpr int PRE = CB == 2 ? TPW : (WOQ_XQM_PRE < TPW ? WOQ_XQM_PRE : TPW);
  auto request = [&](int t, int soff) {  // CB requests
    const bool have = t < cnt && soff != XQM_OOB;  // tiles past the wave's slice / past the last strip: zeros, no traffic
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int s_t = have ? soff + t * 1024 + cb * cb_step : XQM_OOB;
      XQM_LD128_NT(w[cb][t], v16, rq, s_t);
    }
  };
  const int so0 = strip_off(0);
#pragma unroll
  for (int t = 0; t < PRE; ++t) request(t, so0);

  // ---- 2. once per wave: the limb blocks and block factors of its K slice — a few coalesced requests (every workgroup
  // of the launch reads the same L2-resident vector: one lane-per-row request per MFMA operand, the first build of this
  // kernel, was 6x the requests for the same bytes and cost 0.5-1.7 us per launch), parked in wave-private LDS and
  // read from there into registers ONCE for all strips ----
  constexpr int XP = (TPW * 384 + 1023) / 1024;  // 1-KiB pieces per limb strip
  constexpr int UL = (TPW * 8 + 63) / 64;        // block factors per lane (8 blocks per tile)
  const int nblk = uni(max(0, min(cnt, tiles_k - kt0)) * 8);
  const xqm_rsrc rl = xqm_make_rsrc(xlimbs + (size_t)kt0 * 384, nblk * 48);
  const xqm_rsrc ru = xqm_make_rsrc(xu + (size_t)kt0 * 8, nblk * 4);
  const xqm_rsrc rsx = xqm_make_rsrc(xsx + (size_t)kt0 * 8, ASYM ? nblk * 4 : 0);
  u32x4 xl[XP];
  float uw[UL], sxw[UL];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int vo = v16 + j * 1024 < TPW * 384 ? v16 + j * 1024 : XQM_OOB;
    XQM_LD128(xl[j], vo, rl, 0);
  }
#pragma unroll
  for (int j = 0; j < UL; ++j) {
    const int vo = (lane + 64 * j) * 4;
    XQM_LD32(uw[j], vo, ru, 0);
    if constexpr (ASYM)
      XQM_LD32(sxw[j], vo, rsx, 0);
    else
      sxw[j] = 0.f;
  }
  constexpr int NA = XP + UL * (ASYM ? 2 : 1);  // requests of this stage
  // scale (and zero-point) slices of a strip: one vector request per KiB, lanes past the slice request nothing
  constexpr int NSP = (L::SCB + 1023) / 1024;
  constexpr int NSC = CB * (NSP + (ASYM ? 1 : 0));  // requests per strip
  int g0 = 0;
  if constexpr (SMODE == 0) g0 = uni(min(kt0 >> tpg_shift, n_groups - 1));
  const int n_tn = n_strips * CB;
  const xqm_rsrc rs = xqm_make_rsrc(scales, SMODE == 0 ? n_tn * n_groups * 16 * ESZ : n_tn * tiles_k * 64 * ESZ);
  const xqm_rsrc rz = xqm_make_rsrc(zp, ASYM ? (SMODE == 0 ? n_tn * n_groups * 16 : n_tn * tiles_k * 64) : 0);
  u32x4 sl[CB][NSP], zl[CB];
  int vsc[NSP];
#pragma unroll
  for (int j = 0; j < NSP; ++j) vsc[j] = v16 + j * 1024 < L::SCB ? v16 + j * 1024 : XQM_OOB;
  const int vzp = v16 < L::ZPB ? v16 : XQM_OOB;
  auto request_scales = [&](int s) {  // NSC requests
    const bool have = s < ns;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int tn = (b + s * n_wg) * CB + cb;
      const int so_s = have ? uni(SMODE == 0 ? (tn * n_groups + g0) * 16 * ESZ : (tn * tiles_k + kt0) * 64 * ESZ) : XQM_OOB;
#pragma unroll
      for (int j = 0; j < NSP; ++j) XQM_LD128(sl[cb][j], vsc[j], rs, so_s);
      if constexpr (ASYM) {
        const int so_z = have ? uni(SMODE == 0 ? (tn * n_groups + g0) * 16 : (tn * tiles_k + kt0) * 64) : XQM_OOB;
        XQM_LD128(zl[cb], vzp, rz, so_z);
      }
    }
  };
  request_scales(0);
#pragma unroll
  for (int t = PRE; t < TPW; ++t) request(t, so0);
  WOQ_XQS_STAMP(1);
  // the counters are zero before any wave can bump one: LDS only, the window stays in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (wid == 0) {  // park the epilogue inputs: the OLDEST requests of this wave, everything behind them stays in flight
    XQM_WAIT(TPW * CB + NA + NSC);
#pragma unroll
    for (int j = 0; j < 4; ++j) XQM_TOUCH(ssq_v[j]);
    XQM_TOUCH(e_res);
    XQM_TOUCH(g_next);
    if (kq < ns) {
      epi[kq * 32 + i16] = e_res;
      epi[kq * 32 + 16 + i16] = next_norm_w != nullptr ? g_next : 1.f;
    }
    const float4_t t4 = (ssq_v[0] + ssq_v[1]) + (ssq_v[2] + ssq_v[3]);
    const float sq = wave_sum_dpp((t4.x + t4.y) + (t4.z + t4.w));
    if (lane == 0) red[0] = sq;
  }
  // limb blocks, block factors, strip 0's scales: everything older than the last TPW - PRE tile requests. Park them
  // (wave-private LDS, in program order: no workgroup barrier), then the A operand and the factors into registers.
  XQM_WAIT((TPW - PRE) * CB);
  ((uint32_t*)zero_blk)[lane] = 0u;  // every wave writes the same zeros before it reads them
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    XQM_TOUCH(xl[j]);
    if (v16 + j * 1024 < L::STRIP) *(u32x4*)(wbase + v16 + j * 1024) = xl[j];  // past the slice: zeros
  }
#pragma unroll
  for (int j = 0; j < UL; ++j) {
    XQM_TOUCH(uw[j]);
    ((float*)(wbase + L::O_U))[lane + 64 * j] = uw[j];
    if constexpr (ASYM) {
      XQM_TOUCH(sxw[j]);
      ((float*)(wbase + L::O_SX))[lane + 64 * j] = sxw[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  // MFMA row r = lane & 15 -> quarter e = r >> 2, digit p = r & 3 (row 4 e + 3 stays zero); a row is live in lane
  // quarter kq == e only, everything else reads the zero block
  const bool a_live = (i16 >> 2) == kq && (i16 & 3) != 3;
  const unsigned char* a_base = a_live ? wbase + kq * 48 + (i16 & 3) * 16 : zero_blk + kq * 16;
  const int a_step_t = a_live ? 384 : 0, a_step_h = a_live ? 192 : 0;
  i32x4 a[TPW][2];
  float u[TPW][2], sx[TPW][2];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a[t][h] = *(const i32x4*)(a_base + t * a_step_t + h * a_step_h);
      u[t][h] = ((const float*)(wbase + L::O_U))[t * 8 + h * 4 + kq];
      sx[t][h] = ASYM ? ((const float*)(wbase + L::O_SX))[t * 8 + h * 4 + kq] : 0.f;
    }

  const bool bf = (flags & 1) != 0;
  const i32x4 izero = {0, 0, 0, 0};
  const unsigned char* scp0 = wbase + L::O_SC;
  const unsigned char* zpp0 = wbase + L::O_ZP;

  // ---- 3. strips ----
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    // this strip's scale slices: requested one strip ago, in front of this strip's TPW x CB tiles; parked now, the next
    // strip's follow at once. Wave-private LDS in program order: no barrier.
    if (s > 0) XQM_WAIT(TPW * CB);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
      for (int j = 0; j < NSP; ++j) {
        XQM_TOUCH(sl[cb][j]);
        if (v16 + j * 1024 < L::SCB) *(u32x4*)(wbase + L::O_SC + cb * L::SCB + v16 + j * 1024) = sl[cb][j];
      }
      if constexpr (ASYM) {
        XQM_TOUCH(zl[cb]);
        if (v16 < L::ZPB) *(u32x4*)(wbase + L::O_ZP + cb * L::ZPB + v16) = zl[cb];
      }
    }
    const bool more = s + 1 < SMAX;  // compile-time after unrolling
    if (more) request_scales(s + 1);
    const int so_next = more ? strip_off(s + 1) : XQM_OOB;
    float tot[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) tot[cb] = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      // tile t of this strip: behind it sit the rest of this strip's tiles, the next strip's scales and the next
      // strip's tiles 0 .. t - 1 — (TPW - 1) CB + NSC requests whatever t is; on the last strip only its own rest
      if (more)
        XQM_WAIT((TPW - 1) * CB + NSC);
      else
        XQM_WAIT((TPW - 1 - t) * CB);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) XQM_TOUCH(w[cb][t]);
      const float u0 = u[t][0], u1 = u[t][1];
      const float s0 = sx[t][0], s1 = sx[t][1];
      int gi = t;
      if constexpr (SMODE == 0) gi = min((kt0 + t) >> tpg_shift, n_groups - 1) - g0;
      float part[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const u32x4 wv = w[cb][t];
        const unsigned char* scp = scp0 + cb * L::SCB;
        const unsigned char* zpp = zpp0 + cb * L::ZPB;
        const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u),
                          (int)((wv.y << 4) & 0xf0f0f0f0u), (int)(wv.y & 0xf0f0f0f0u)};
        const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u),
                          (int)((wv.w << 4) & 0xf0f0f0f0u), (int)(wv.w & 0xf0f0f0f0u)};
        float f0 = digit_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][0], b0, izero, 0, 0, 0));
        float f1 = digit_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][1], b1, izero, 0, 0, 0));
        if constexpr (SMODE == 0) {
          if constexpr (ASYM) {  // the weights carry 16 * q: the zero point enters as 16 * zp
            const float z16 = -16.f * (float)((int)zpp[gi * 16 + i16] - 8);
            f0 = fmaf(z16, s0, f0);
            f1 = fmaf(z16, s1, f1);
          }
          float sc;
          if constexpr (S32)
            sc = *(const float*)(scp + (gi * 16 + i16) * 4);
          else
            sc = tscale16(*(const uint16_t*)(scp + (gi * 16 + i16) * 2), bf);
          part[cb] = sc * fmaf(f0, u0, f1 * u1);
        } else {  // this lane quarter's 32-k group of each half: s = 2 h + (kq >> 1)
          const int o = (t * 16 + i16) * 4 + (kq >> 1);
          if constexpr (ASYM) {
            f0 = fmaf(-16.f * (float)((int)zpp[o] - 8), s0, f0);
            f1 = fmaf(-16.f * (float)((int)zpp[o + 2] - 8), s1, f1);
          }
          float sc0, sc1;
          if constexpr (S32) {
            sc0 = *(const float*)(scp + o * 4);
            sc1 = *(const float*)(scp + (o + 2) * 4);
          } else {
            sc0 = tscale16(*(const uint16_t*)(scp + o * 2), bf);
            sc1 = tscale16(*(const uint16_t*)(scp + (o + 2) * 2), bf);
          }
          part[cb] = fmaf(sc0 * u0, f0, (sc1 * u1) * f1);
        }
      }
      // the slot is free once the B operands are built: tile t of the next strip. The request carries a dependency on
      // this tile's result so that it cannot be hoisted above the reads of the registers it overwrites.
      if (more) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) XQM_TOUCH(part[cb]);
        request(t, so_next);
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) tot[cb] += part[cb];
      if (s == 0 && t == 0) WOQ_XQS_STAMP(2);
    }
    if (s == SMAX - 1) WOQ_XQS_STAMP(3);
    if (s >= ns) continue;  // a strip this workgroup does not have: its requests moved no bytes, its sums are zeros
    // the four lane quarters hold the four blocks' shares of each column
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      float v = tot[cb];
      v += xqs_swap32(v);
      v += xqs_swap16(v);
      if (lane < 16) slab[(((size_t)s * nw + wid) * CB + cb) * 16 + lane] = v;
    }
    asm volatile("" ::: "memory");
    unsigned int arrived = 0u;
    if (lane == 0) arrived = lds_inc(cnts + s);
    asm volatile("" ::: "memory");
    if (uni((int)arrived) != nw - 1) continue;

    // ---- 4. finish strip s (its last wave, lanes 0..15): sum over waves, RMSNorm factor, bias, SiLU*mul, residual ----
    if (s == SMAX - 1) WOQ_XQS_STAMP(4);
    if (lane < 16) {
      const int sj = b + s * n_wg;
      float v = 0.f, up = 0.f;
#pragma unroll 4
      for (int w2 = 0; w2 < nw; ++w2) {
        v += slab[(((size_t)s * nw + w2) * CB) * 16 + lane];
        if constexpr (CB == 2) up += slab[(((size_t)s * nw + w2) * CB + 1) * 16 + lane];
      }
      const float inv = ssq_in != nullptr ? 1.0f / sqrtf(red[0] / (float)K + eps) : 1.f;  // HF LlamaRMSNorm
      v *= inv;
      const int n = sj * 16 + lane;  // CB == 1 or the SiLU pair: one 16-column output tile per strip
      if (silu) {
        up *= inv;
        if (bias) {
          v += bias[min((sj * 2) * 16 + lane, N - 1)];
          up += bias[min((sj * 2 + 1) * 16 + lane, N - 1)];
        }
        v = v / (1.0f + __expf(-v)) * up;
      } else if (bias) {
        v += bias[min(n, N - 1)];
      }
      const bool live = n < n_out;
      v = live ? v + epi[s * 32 + lane] : 0.f;
      if (tp != nullptr && live) {  // row-parallel projection under tensor parallelism: push the partial (woq_gemv_xqs.h)
        const unsigned int seq = tp->ctl[0];
        const int world = tp->world, rank = tp->rank;
        for (int r = 0; r < world; ++r)
          if (r != rank) push(tp->peer[r] + ar_slot(*tp, (int)(seq & 1u), rank, (unsigned int)n), __float_as_uint(v), seq);
      }
      if (live && out) {
        if constexpr (FUSED)
          __hip_atomic_store((unsigned long long*)out + n, ((unsigned long long)fused_tag << 32) | __float_as_uint(v),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          out[n] = v;
      }
      if (xo.limbs != nullptr) {  // this strip IS block sj of the next kernel's activation vector
        const float ss = ssq_out != nullptr ? row16_sum(v * v) : 0.f;
        xq_emit16<false>(v * epi[s * 32 + 16 + lane], xo, sj, lane, XqPub{nullptr, 0u}, ssq_out, ss);
      }
    }
    if (s == SMAX - 1) WOQ_XQS_STAMP(5);
  }
}

// WMAX: waves per workgroup the instantiation admits (8 | 12 | 16 -> 256 | 168 | 128 registers per lane): the A operand
// takes 8 registers per tile of the wave's slice, the window 4 per tile and column tile.
// SMAX: strips per workgroup, unrolled (a workgroup with fewer runs the rest on zeros: no traffic, no epilogue).
template <int TPW, int CB, int SMODE, bool ASYM, bool S32, int SMAX, int WMAX>
__global__ __launch_bounds__(WMAX * 64) void gemv_xqm_kernel(
    const u32x4* __restrict__ q, const void* __restrict__ scales, const uint8_t* __restrict__ xlimbs,
    const float* __restrict__ xu, int tiles_k, int kt_off, int base_tiles, int rem_tiles, int n_groups, int tpg_shift,
    const uint8_t* __restrict__ zp, const float* __restrict__ xsx, float* __restrict__ out,
    const float* __restrict__ bias, const float* residual, float eps, int N, int K, int flags,
    const float* __restrict__ ssq_in, int n_ssq, XqPtrs xo, const float* __restrict__ next_norm_w,
    float* __restrict__ ssq_out, const CommDev* __restrict__ tp, int n_strips, int n_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  gemv_xqm_body<TPW, CB, SMODE, ASYM, S32, false, SMAX>(smem_raw, q, scales, xlimbs, xu, tiles_k, kt_off, base_tiles,
                                                        rem_tiles, n_groups, tpg_shift, zp, xsx, out, bias, residual, eps,
                                                        N, K, flags, ssq_in, n_ssq, xo, next_norm_w, ssq_out, tp,
                                                        n_strips, n_wg, 0u);
}

}  // namespace woq
