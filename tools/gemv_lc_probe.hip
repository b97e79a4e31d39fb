// woq_gemv_lc.hip — batch-1 decode GEMV as ONE LOADER WAVE + SEVEN CONSUMER WAVES per CU (XQ activations in, XQ out).
//
// Why another decode kernel. In gemv_tile_kernel / gemv_xq_kernel every wave issues its own weight loads and then
// computes on them. A CU's vector-memory path holds ~16 KiB of misses, so a wave that has more than that to request
// sits in the ISSUE of its loads until most of its data has already arrived — and only then starts its eight tiles of
// unpack / MFMA / recombination, back to back, after the stream is over. Knock-outs of the kernel's stages
// (profiles/r02b_gemv_knockouts.txt) put 2.7-3.3 us of every 5-12 us launch into work that runs after, not under, the
// weight stream; the same launches with loads only take 3.3 / 5.0 / 7.9 / 5.0 us.
// Here the stream and the arithmetic never share a wave:
//   * wave 0 (loader) streams the workgroup's weight tiles HBM -> LDS ring with LDS-DMA (global_load_lds_dwordx4 nt,
//     1 KiB per instruction, no VGPRs), up to 32 pieces in flight, and publishes how many have landed
//     (s_waitcnt vmcnt(31) after every issue -> one LDS counter). A workgroup's tiles are one contiguous byte range
//     of the blob (strips are contiguous along K, adjacent strips are adjacent), so the loader's address is a pointer
//     that advances by 1 KiB;
//   * waves 1..7 (consumers) take tiles round-robin: poll the counter, ds_read_b128 their 16 B of the tile, A
//     operands from the XQ limb blocks staged once per workgroup, 2 x v_mfma_i32_16x16x64_i8, per-block
//     recombination (exactly gemv_xq_kernel's arithmetic, woq_gemv_xq.hip), partial sums per strip into an LDS slab;
//   * one workgroup per CU, each owning a contiguous run of column strips (pairs for the fused gate/up SiLU*mul);
//     ring slots are reused once the consumer that owned the previous tenant reports it done (one LDS progress word
//     per consumer, checked by the loader before it issues into a used slot; never needed when the workgroup's tiles
//     fit the ring: up to 128 KiB);
//   * epilogue as gemv_xq_kernel: RMSNorm factor from the producer's partial sums of squares, bias, SiLU*mul,
//     residual, fp32 store, and the result as the next kernel's XQ vector.
// Every spin is bounded: a consumer that never sees its tile sets `err` and carries on (the launch finishes, the
// engine reports the error) — it cannot hang the GPU.
#include <algorithm>

#include "woq_gemv_common.h"
#include "woq_xq.h"

namespace woq {

constexpr int LC_NC = 15;           // consumer waves
constexpr int LC_THREADS = 64 * (1 + LC_NC);
#ifndef WOQ_LC_D
#define WOQ_LC_D 32
#endif
constexpr int LC_D = WOQ_LC_D;      // LDS-DMA pieces in flight, a multiple of 4 (vmcnt is a 6-bit counter)
#ifdef WOQ_LC_NO_NT
#define WOQ_LC_NT ""
#else
#define WOQ_LC_NT " nt"
#endif
constexpr int LC_SPIN_LIMIT = 1 << 22;

struct LcArgs {
  const uint8_t* q;  // qdata of the blob: [strip][tiles_k][1 KiB]
  const void* scales;
  const uint8_t* zp;
  const uint8_t* xlimbs;
  const float* xu;
  const float* xsx;
  int tiles_k, kt_begin, kt_count;  // K tiles of the blob, and the range this launch covers
  int K, n_groups, tpg_shift;
  int units;  // column units: strips, or gate/up strip pairs
  int N, flags;
  float* out;
  const float* bias;
  const float* residual;
  float eps;
  const float* ssq_in;
  int n_ssq;
  XqPtrs xo;
  const float* next_norm_w;
  float* ssq_out;
  int sc_strip, zp_strip;  // bytes of one strip's scale / zero-point array (zp_strip 0 = symmetric)
  int ring_tiles;          // R >= 2 * LC_D slots of 1 KiB
  int max_units;   // per workgroup
  uint32_t* err;
};

// LDS (bytes): [ctl 256: landed, staged, red...][limbs kt_count * 384][u kt_count * 32][sx kt_count * 32]
//              [scales max_units * CB strips x sc_strip][zero points ... x zp_strip][zero 64][ones 64]
//              [slab NC x max_units x CB x 16 f32][ring R KiB]
// sc_strip / zp_strip: bytes of one strip's scale / zero-point array in the blob (all K groups of the strip)
__host__ __device__ inline size_t lc_fixed_bytes(int kt_count, int max_units, int cb, int sc_strip, int zp_strip) {
  size_t b = 256 + (size_t)kt_count * 384 + (size_t)kt_count * 64 + (size_t)max_units * cb * (sc_strip + zp_strip) +
             128 + (size_t)LC_NC * max_units * cb * 64;
  return (b + 1023) & ~(size_t)1023;
}

// one 1-KiB LDS-DMA piece: lane l moves 16 B from gsrc (per lane) to lds_dst + 16 l. M0 carries the LDS address and
// is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md §5.7). The load is invisible
// to hipcc's waitcnt bookkeeping — every wait on it below is explicit.
__device__ __forceinline__ void lds_dma_1k(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" WOQ_LC_NT "\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// four pieces that are contiguous in memory AND in the ring: one M0 / address set-up, the instruction offset advances
// both addresses (cdna_hip_programming.md §5.7; the prefill GEMM's A-operand mover uses the same form)
__device__ __forceinline__ void lds_dma_4k(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off" WOQ_LC_NT "\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024" WOQ_LC_NT "\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048" WOQ_LC_NT "\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072" WOQ_LC_NT "\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int N, typename F>
__device__ __forceinline__ void lc_drain(F&& landed_all_but) {  // vmcnt(N), vmcnt(N - 4), ..., vmcnt(0)
  wait_vmcnt<N>();
  landed_all_but(N);
  if constexpr (N >= 4) lc_drain<N - 4>(landed_all_but);
}

template <int CB, int SMODE, bool ASYM, bool S32>
__global__ __launch_bounds__(LC_THREADS) void gemv_lc_kernel(LcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int u0 = (int)(((long long)b * a.units) / G), u1 = (int)(((long long)(b + 1) * a.units) / G);
  const int nu = u1 - u0;                      // column units of this workgroup
  const int T = nu * CB * a.kt_count;          // its weight tiles, in (unit, cb, k) order
  const int R = a.ring_tiles;
  // [0] landed tiles, [1] consumers done staging, [2] RMSNorm sum, [8 + c] tiles consumer c is done with (index + 1)
  // (an explicit LDS pointer: through a generic `volatile int*` hipcc emits flat accesses followed by vmcnt(0),
  // which in the loader would drain the DMA queue at every publish)
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  lds_vint* ctl = (lds_vint*)(__attribute__((address_space(3))) void*)smem_raw;
  unsigned char* limbs = smem_raw + 256;
  float* lu = (float*)(limbs + (size_t)a.kt_count * 384);
  float* lsx = lu + (size_t)a.kt_count * 8;
  unsigned char* lsc = (unsigned char*)(lsx + (size_t)a.kt_count * 8);  // this workgroup's strips' scales, as in the blob
  unsigned char* lzp = lsc + (size_t)a.max_units * CB * a.sc_strip;
  unsigned char* zero_blk = lzp + (size_t)a.max_units * CB * a.zp_strip;
  unsigned char* ones_blk = zero_blk + 64;
  float* slab = (float*)(ones_blk + 64);  // [NC][max_units * CB][16]
  const uint32_t ring_off = (uint32_t)lc_fixed_bytes(a.kt_count, a.max_units, CB, a.sc_strip, a.zp_strip);
  unsigned char* ring = smem_raw + ring_off;
  const bool silu = (a.flags & 2) != 0, bf = (a.flags & 1) != 0;
  // timing experiments (WOQ_LC_DEBUG): 1 consumers skip the arithmetic, 2 consumers do not wait for tiles, 4 the loader
  // issues nothing. Results are garbage under any of them.
  const bool dbg_nomath = (a.flags & 0x100) != 0, dbg_nowait = (a.flags & 0x200) != 0, dbg_noload = (a.flags & 0x400) != 0;
  const int n_out = silu ? (a.N >> 1) : a.N;

  // residual element of the epilogue threads (16 per unit), fetched before anything else
  float e_res = 0.f;
  if (a.residual != nullptr && tid < nu * 16) {
    const int n = (u0 + (tid >> 4)) * 16 + (tid & 15);
    if (n < n_out) e_res = a.residual[n];
  }
  // ---- init: counters, constant blocks, slab (a consumer may own no tile of some strip) ----
  if (tid < 16) ctl[tid] = 0;
  if (tid < 16) {
    ((uint32_t*)zero_blk)[tid] = 0u;
    ((uint32_t*)ones_blk)[tid] = 0x01010101u;
  }
  for (int i = tid; i < LC_NC * nu * CB * 16; i += LC_THREADS) slab[i] = 0.f;
  __syncthreads();

  if (wid == 0) {
    // ================================ loader ================================
    const int strip0 = u0 * CB;
    const uint8_t* src = a.q + ((size_t)strip0 * a.tiles_k + a.kt_begin) * 1024 + lane * 16;
    const size_t strip_gap = (size_t)(a.tiles_k - a.kt_count) * 1024;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)ring;
    int ktl = 0, slot = 0, j = 0, pub = 0;  // pub: landed count published so far (never decreases)
    bool ok = true;
    auto publish = [&](int n) {
      if (n > pub) {
        pub = n;
        if (lane == 0) ctl[0] = n;
      }
    };
    // fast path: bursts of four tiles (the loop body must stay well under the ~100 clk a CU takes to land 1 KiB;
    // one tile per iteration measured 2x slower than the every-wave-loads kernel). Needs the workgroup's tiles
    // contiguous in memory (no K-chunk gap) and a ring of a multiple of four slots.
    if (dbg_noload) {
      j = T;
      publish(T);
    }
    if (strip_gap == 0 && (R & 3) == 0) {
      for (; j + 4 <= T; j += 4) {
        if (j + 4 > R) {  // the four slots' previous tenants, tiles j - R .. j - R + 3, must have been consumed
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int prev = j + q4 - R;
            if (prev < 0) continue;
            const int owner = prev % LC_NC;
            for (int spins = 0; ok && ctl[8 + owner] <= prev; ++spins) {  // once a wait gave up, none is repeated
              if (spins > LC_SPIN_LIMIT) ok = false;
              __builtin_amdgcn_s_sleep(1);
            }
          }
        }
        lds_dma_4k(src, __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)slot * 1024u));
        src += 4096;
        slot += 4;
        if (slot >= R) slot = 0;
        wait_vmcnt<LC_D - 4>();  // at most 28 pieces still in flight
        publish(j + 4 - (LC_D - 4));
      }
    }
    for (; j < T; ++j) {  // tail, and the general path (K-chunk gaps between strips)
      if (j >= R) {
        const int prev = j - R, owner = prev % LC_NC;
        for (int spins = 0; ok && ctl[8 + owner] <= prev; ++spins) {
          if (spins > LC_SPIN_LIMIT) ok = false;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      lds_dma_1k(src, __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)slot * 1024u));
      src += 1024;
      if (++ktl == a.kt_count) {
        ktl = 0;
        src += strip_gap;
      }
      if (++slot == R) slot = 0;
      wait_vmcnt<LC_D - 1>();  // at most 31 pieces still in flight: piece j - 31 has landed
      publish(j - (LC_D - 2));
    }
    // drain: publish the tail as it lands, four pieces at a time
    lc_drain<LC_D - 4>([&](int left) { publish(T - left); });
    if (!ok && lane == 0) atomicOr(a.err, 2u);
  } else {
    // ================================ consumers ================================
    const int c = wid - 1;
    const int i16 = lane & 15, kq = lane >> 4;
    {  // stage the XQ input of this launch's K range: limb blocks, block factors (and block sums)
      const int limb_bytes = a.kt_count * 384;
      const uint8_t* lsrc = a.xlimbs + (size_t)a.kt_begin * 384;
      for (int off = (c * 64 + lane) * 16; off < limb_bytes; off += LC_NC * 1024)
        *(u32x4*)(limbs + off) = *(const u32x4*)(lsrc + off);
      const int nb = a.kt_count * 8;
      for (int i = c * 64 + lane; i < nb; i += LC_NC * 64) {
        lu[i] = a.xu[(size_t)a.kt_begin * 8 + i];
        if constexpr (ASYM) lsx[i] = a.xsx[(size_t)a.kt_begin * 8 + i];
      }
      // the strips' scales (and zero points): one contiguous range of the blob each. They are requested NOW, at the
      // head of the CU's memory queue — a per-tile request later would sit behind the loader's stream every time
      // (measured: one-tile-ahead scale loads made this kernel 1.6x slower than the every-wave-loads one)
      const int sc_bytes = nu * CB * a.sc_strip;
      const uint8_t* ssrc = (const uint8_t*)a.scales + (size_t)u0 * CB * a.sc_strip;
      for (int off = (c * 64 + lane) * 16; off < sc_bytes; off += LC_NC * 1024)
        *(u32x4*)(lsc + off) = *(const u32x4*)(ssrc + off);
      if constexpr (ASYM) {
        const int zp_bytes = nu * CB * a.zp_strip;
        const uint8_t* zsrc = a.zp + (size_t)u0 * CB * a.zp_strip;
        for (int off = (c * 64 + lane) * 16; off < zp_bytes; off += LC_NC * 1024)
          *(u32x4*)(lzp + off) = *(const u32x4*)(zsrc + off);
      }
    }
    float4_t ssq_v[4];
    if (a.ssq_in != nullptr && c == 0) {
      const rsrc_t rs = make_rsrc(a.ssq_in, a.n_ssq * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        ssq_v[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + j * 1024, 0, 0));
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)
      __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)&ctl[1], 1, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
    bool ok = true;
    for (int spins = 0; ok && ctl[1] < LC_NC; ++spins) {  // all consumers' pieces of the activation vector are in LDS
      if (spins > LC_SPIN_LIMIT) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    // A-operand addresses: MFMA row r = lane & 15 -> lane quarter e = r >> 2, part r & 3 (limb 0..2 | ones)
    const int a_e = i16 >> 2, a_part = i16 & 3;
    const bool a_live = a_e == kq;
    const unsigned char* a_base =
        !a_live ? zero_blk + kq * 16 : (a_part == 3 ? ones_blk + kq * 16 : limbs + kq * 48 + a_part * 16);
    const int a_step_t = (a_live && a_part != 3) ? 384 : 0, a_step_h = (a_live && a_part != 3) ? 192 : 0;
    const i32x4 izero = {0, 0, 0, 0};

    // scale / zero point words of one tile for this lane, from the staged arrays; (sidx, ktl) -> the workgroup's
    // strip sidx, K tile kt_begin + ktl
    typedef typename RawSc<SMODE, S32>::type sc_t;
    auto load_sc = [&](int sidx, int ktl, sc_t& sc, uint32_t& z) {
      const int kt = a.kt_begin + ktl;
      if constexpr (SMODE == 0) {
        const int grp = min(kt >> a.tpg_shift, a.n_groups - 1);
        const size_t si = (size_t)grp * 16 + i16;
        if constexpr (S32)
          sc = ((const float*)(lsc + (size_t)sidx * a.sc_strip))[si];
        else
          sc = ((const uint16_t*)(lsc + (size_t)sidx * a.sc_strip))[si];
        if constexpr (ASYM) z = (lzp + (size_t)sidx * a.zp_strip)[si];
      } else {
        const size_t si = (size_t)kt * 16 + i16;  // 4 x 32-k groups per (tile, column)
        if constexpr (S32)
          sc = ((const float4_t*)(lsc + (size_t)sidx * a.sc_strip))[si];
        else
          sc = ((const uint2*)(lsc + (size_t)sidx * a.sc_strip))[si];
        if constexpr (ASYM) z = ((const uint32_t*)(lzp + (size_t)sidx * a.zp_strip))[si];
      }
    };

    int j = c, sidx = 0, ktl = c, slot = c % R;
    while (ktl >= a.kt_count) {
      ktl -= a.kt_count;
      ++sidx;
    }
    float acc = 0.f;
    int acc_sidx = sidx;
    auto flush = [&]() {  // the four lane quarters hold the four blocks' shares of each column
      float v = acc;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) slab[((size_t)c * a.max_units * CB + acc_sidx) * 16 + lane] = v;
      acc = 0.f;
    };
    for (; j < T; j += LC_NC) {
      if (sidx != acc_sidx) {
        flush();
        acc_sidx = sidx;
      }
      sc_t sc;
      uint32_t zw = 0;
      load_sc(sidx, ktl, sc, zw);
      const int t_here = ktl;
      ktl += LC_NC;  // this consumer's next tile
      while (ktl >= a.kt_count) {
        ktl -= a.kt_count;
        ++sidx;
      }
      for (int spins = 0; ok && !dbg_nowait && ctl[0] <= j; ++spins) {  // tile j has landed in the ring
        if (spins > LC_SPIN_LIMIT) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (dbg_nomath) {
        if (lane == 0) ctl[8 + c] = j + 1;
        continue;
      }
      const u32x4 wv = *(const u32x4*)(ring + (size_t)slot * 1024 + lane * 16);
      slot += LC_NC;
      if (slot >= R) slot -= R;
      const i32x4 a0 = *(const i32x4*)(a_base + t_here * a_step_t);
      const i32x4 a1 = *(const i32x4*)(a_base + t_here * a_step_t + a_step_h);
      const float u0f = lu[t_here * 8 + kq], u1f = lu[t_here * 8 + 4 + kq];
      const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u), (int)((wv.y << 4) & 0xf0f0f0f0u),
                        (int)(wv.y & 0xf0f0f0f0u)};
      const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u), (int)((wv.w << 4) & 0xf0f0f0f0u),
                        (int)(wv.w & 0xf0f0f0f0u)};
      const i32x4 d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, izero, 0, 0, 0);
      const i32x4 d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, izero, 0, 0, 0);
      float f0 = limb_combine(d0), f1 = limb_combine(d1);
      if constexpr (SMODE == 0) {
        if constexpr (ASYM) {  // the weights carry 16 * q: the zero point enters as 16 * zp
          const float z16 = -16.f * (float)((int)(zw & 0xff) - 8);
          f0 = fmaf(z16, lsx[t_here * 8 + kq], f0);
          f1 = fmaf(z16, lsx[t_here * 8 + 4 + kq], f1);
        }
        float s;
        if constexpr (S32)
          s = sc;
        else
          s = tscale16(sc, bf);
        acc = fmaf(s, fmaf(f0, u0f, f1 * u1f), acc);
      } else {  // this lane quarter's 32-k group of each half: 2 h + (kq >> 1)
        if constexpr (ASYM) {
          const uint32_t z = zw >> (8 * (kq >> 1));
          f0 = fmaf(-16.f * (float)((int)(z & 0xffu) - 8), lsx[t_here * 8 + kq], f0);
          f1 = fmaf(-16.f * (float)((int)((z >> 16) & 0xffu) - 8), lsx[t_here * 8 + 4 + kq], f1);
        }
        float s0, s1;
        if constexpr (S32) {
          s0 = (kq >> 1) ? sc.y : sc.x;
          s1 = (kq >> 1) ? sc.w : sc.z;
        } else {
          s0 = tscale16((kq >> 1) ? (sc.x >> 16) : (sc.x & 0xffffu), bf);
          s1 = tscale16((kq >> 1) ? (sc.y >> 16) : (sc.y & 0xffffu), bf);
        }
        acc = fmaf(s0 * u0f, f0, fmaf(s1 * u1f, f1, acc));
      }
      // the tile's bytes are in registers (the MFMAs above consumed them): its ring slot may be refilled
      if (lane == 0) ctl[8 + c] = j + 1;
    }
    if (c < T) flush();  // this consumer had at least one tile
    if (a.ssq_in != nullptr && c == 0) {
      const float4_t t4 = (ssq_v[0] + ssq_v[1]) + (ssq_v[2] + ssq_v[3]);
      const float s = wave_sum_dpp((t4.x + t4.y) + (t4.z + t4.w));
      if (lane == 0) ctl[2] = __float_as_int(s);
    }
    if (!ok && lane == 0) atomicOr(a.err, 1u);
  }
  __syncthreads();

  // ---- finish: 16 threads per column unit ----
  if (tid < nu * 16) {
    const int lu_i = tid >> 4, i = tid & 15;
    float v = 0.f, up = 0.f;
#pragma unroll
    for (int c = 0; c < LC_NC; ++c) {
      v += slab[((size_t)c * a.max_units * CB + lu_i * CB) * 16 + i];
      if constexpr (CB == 2) up += slab[((size_t)c * a.max_units * CB + lu_i * CB + 1) * 16 + i];
    }
    const float inv = a.ssq_in != nullptr ? 1.0f / sqrtf(__int_as_float(ctl[2]) / (float)a.K + a.eps) : 1.f;
    v *= inv;
    const int unit = u0 + lu_i, n = unit * 16 + i;
    if (silu) {
      up *= inv;
      if (a.bias) {
        v += a.bias[min((unit * 2) * 16 + i, a.N - 1)];
        up += a.bias[min((unit * 2 + 1) * 16 + i, a.N - 1)];
      }
      v = v / (1.0f + __expf(-v)) * up;
    } else if (a.bias) {
      v += a.bias[min(n, a.N - 1)];
    }
    const bool live = n < n_out;
    v = live ? v + e_res : 0.f;
    if (live && a.out) a.out[n] = v;
    if (a.xo.limbs != nullptr) {  // unit `unit` IS block `unit` of the next kernel's activation vector
      if (a.ssq_out != nullptr) {
        const float ss = row16_sum(v * v);
        if (i == 0) a.ssq_out[unit] = ss;
      }
      const float g = a.next_norm_w != nullptr ? a.next_norm_w[min(n, n_out - 1)] : 1.f;
      xq_emit16(v * g, a.xo, unit, i);
    }
  }
}

template <int CB>
static int launch_lc_sm(const LcArgs& a, int smode, bool asym, bool s32, int grid, size_t lds, hipStream_t st) {
#define WOQ_LC_CASE(SM, AS, S3)                                                                                  \
  if (smode == SM && asym == AS && s32 == S3) {                                                                  \
    auto kern = gemv_lc_kernel<CB, SM, AS, S3>;                                                                  \
    static bool attr_set = false;                                                                                \
    if (!attr_set) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e)); \
      attr_set = true;                                                                                           \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LC_THREADS), lds, st, a);                                          \
    return 0;                                                                                                    \
  }
  WOQ_LC_CASE(0, false, false)
  WOQ_LC_CASE(0, false, true)
  WOQ_LC_CASE(0, true, false)
  WOQ_LC_CASE(0, true, true)
  WOQ_LC_CASE(1, false, false)
  WOQ_LC_CASE(1, false, true)
  WOQ_LC_CASE(1, true, false)
  WOQ_LC_CASE(1, true, true)
#undef WOQ_LC_CASE
  return woq::fail("QBits: bad loader/consumer GEMV configuration");
}

static int lc_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}

// K tiles per launch: the activation vector's limb blocks must fit LDS next to a ring of at least 2 x LC_D tiles
static constexpr int LC_MAX_KT = 160;  // K 20480: 60 KiB of limbs, 10 KiB of factors

bool gemv_lc_supported(const woq_blob_header& h, int epi) {
  if (h.weight_type != WOQ_W_INT4_CLIP || h.off_shuffle != 0 || (h.K % WOQ_TILE_K) != 0 || h.K != h.Kpad) return false;
  if (h.Npad != h.N || (h.N % 16) != 0) return false;
  if (epi == 1 && ((h.Npad / WOQ_TILE_N) & 1)) return false;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return false;
  }
  // LDS: activation blocks + the workgroup's strips' scales must leave room for a ring of 2 x LC_D tiles
  const int cb = epi == 1 ? 2 : 1, tiles_k = h.Kpad / WOQ_TILE_K;
  const int units = (h.Npad / WOQ_TILE_N) / cb, grid = std::min(units, lc_cu_count());
  const int max_units = (units + grid - 1) / grid;
  const int chunks = (tiles_k + LC_MAX_KT - 1) / LC_MAX_KT, per = (tiles_k + chunks - 1) / chunks;
  const int slots = h.scale_mode == 0 ? h.n_groups * 16 : tiles_k * 64;
  const size_t fixed = lc_fixed_bytes(per, max_units, cb, slots * (h.scale_type == WOQ_F32 ? 4 : 2), h.off_zp ? slots : 0);
  return fixed + (size_t)2 * LC_D * 1024 <= 160 * 1024;
}

// Same contract as launch_gemv_xq (woq_gemv_xq.hip). `err` = device word that collects spin timeouts.
int launch_gemv_lc(const XqPtrs& xin, const void* blob, const woq_blob_header& h, const float* bias, float* out,
                   const float* ssq_in, float eps, const float* residual, int epi, const XqPtrs& xo,
                   const float* next_norm_w, float* ssq_out, uint32_t* err, hipStream_t st) {
  if (!gemv_lc_supported(h, epi)) return woq::fail("QBits: shape not covered by the loader/consumer GEMV");
  LcArgs a;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = b + h.off_q;
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.xlimbs = xin.limbs;
  a.xu = xin.u;
  a.xsx = xin.sx;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
  a.K = h.K;
  a.N = h.N;
  a.n_groups = h.n_groups;
  a.tpg_shift = 0;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    int tpg = h.group / WOQ_TILE_K;
    while (tpg > 1) {
      tpg >>= 1;
      ++a.tpg_shift;
    }
  }
  a.flags = (h.scale_type == WOQ_BF16 ? 1 : 0) | (epi == 1 ? 2 : 0);
  static const int dbg = [] {
    const char* v = getenv("WOQ_LC_DEBUG");
    return v ? atoi(v) : 0;
  }();
  a.flags |= (dbg & 7) << 8;
  a.bias = bias;
  a.eps = eps;
  a.ssq_in = ssq_in;
  a.n_ssq = h.K / 16;
  if (ssq_in != nullptr && a.n_ssq > 1024) return woq::fail("QBits: RMSNorm partials beyond K = 16384");
  a.next_norm_w = next_norm_w;
  a.ssq_out = ssq_out;
  a.err = err;
  const int cb = epi == 1 ? 2 : 1;
  a.units = (h.Npad / WOQ_TILE_N) / cb;
  const int grid = std::min(a.units, lc_cu_count());
  a.max_units = (a.units + grid - 1) / grid;
  {
    const int esz = h.scale_type == WOQ_F32 ? 4 : 2;
    const int per_strip = h.scale_mode == 0 ? h.n_groups * 16 : a.tiles_k * 64;  // scale slots of one strip
    a.sc_strip = per_strip * esz;
    a.zp_strip = a.zp ? per_strip : 0;
  }
  const int chunks = (a.tiles_k + LC_MAX_KT - 1) / LC_MAX_KT;
  if (chunks > 1 && (ssq_in != nullptr || out == nullptr || epi != 0))
    return woq::fail("QBits: a K range split over chained launches takes no norm / SiLU and needs an fp32 output");
  const int per = (a.tiles_k + chunks - 1) / chunks;
  for (int c = 0; c < chunks; ++c) {
    a.kt_begin = c * per;
    a.kt_count = std::min(per, a.tiles_k - a.kt_begin);
    if (a.kt_count <= 0) break;
    const bool last = a.kt_begin + a.kt_count >= a.tiles_k;
    const size_t fixed = lc_fixed_bytes(a.kt_count, a.max_units, cb, a.sc_strip, a.zp_strip);
    int ring = (int)((160 * 1024 - fixed) / 1024);
    ring = std::min(128, ring) & ~3;
    if ((int)((160 * 1024 - (long long)fixed) / 1024) < 2 * LC_D)
      return woq::fail("QBits: loader/consumer GEMV: no room for the weight ring");
    a.ring_tiles = ring;
    a.out = out;
    a.bias = c == 0 ? bias : nullptr;
    a.residual = c == 0 ? residual : out;  // chunk c > 0 adds onto the previous chunk's output
    a.xo = last ? xo : XqPtrs{nullptr, nullptr, nullptr};
    const size_t lds = fixed + (size_t)ring * 1024;
    const int smode = (int)h.scale_mode;
    const bool asym = a.zp != nullptr, s32 = h.scale_type == WOQ_F32;
    const int rc = cb == 2 ? launch_lc_sm<2>(a, smode, asym, s32, grid, lds, st)
                           : launch_lc_sm<1>(a, smode, asym, s32, grid, lds, st);
    if (rc) return rc;
  }
  return 0;
}

}  // namespace woq
