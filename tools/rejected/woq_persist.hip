// woq_persist.hip — the decode step's layers as ONE persistent launch: a weight stream that never stops at an operator
// boundary (round 3). Reference path replaced: the 7 qbits.woq_linear calls per layer of QuantizedLinearQBits.forward
// (transformers/llm/quantization/nn/modules.py:140-169 -> qbits.cpp:113-140) with stock HF attention between them.
//
// Why. With one launch per projection the weight stream is rebuilt from nothing 4 times per layer: ~2 us of launch
// boundary, then a ramp before the first tiles land, then a tail while the last waves finish — the step's GEMV launches
// average 0.47 of 8 TB/s and their load-only twins 0.64 (profiles/r03n). Chaining the launches of a layer with
// register prefetch did not help (profiles/r03y): a wave that waits for its input cannot hold more than a few KiB of
// weights in flight. LDS can: tools/dma_stream_probe.hip (profiles/r03aa) — ONE wave per CU issuing
// global_load_lds_dwordx4 streams at the chip's full rate (6.1 TB/s over a layer's 96 MiB, 6.5 TB/s steady state),
// whatever the other waves of the workgroup do.
// How. One workgroup per CU, 12 waves:
//   * wave 0, the LOADER, walks the whole token's weights in operator order — for every projection the contiguous KiB
//     range of this workgroup's column strips — HBM -> LDS ring (LDS-DMA, no VGPRs, up to 32 KiB in flight) and
//     publishes how many tiles have landed. It needs nothing but the weights' addresses, so it runs up to a ring
//     (~100 KiB per CU, ~25 MiB chip-wide, ~4 us of stream) AHEAD of the arithmetic, across operator boundaries;
//   * waves 1..11, the CONSUMERS, take landed tiles round-robin (ds_read_b128, 2 x v_mfma_i32_16x16x64_i8 over the XQ
//     digits of the activation vector, per-block recombination — woq_gemv_xqs.h's arithmetic), leave per-strip partial
//     sums in an LDS slab, and one of them runs the epilogue (RMSNorm factor, SiLU * mul, residual) for the
//     workgroup's 16-column units;
//   * an operator's output vector crosses to all workgroups as TAGGED GRANULES (cdna_hip_programming.md Guideline 16,
//     form R2: 8-byte write-through stores, 16 tag bits + 48 payload bits = two values' three balanced digits; the data
//     is its own flag, no fence and no ordering between stores). Every consumer wave sweeps its share of the vector
//     and re-reads only the granules whose tag is not yet this (step, layer, operator)'s. (Optional hint words — "my
//     blocks are on their way", polled by one wave before the sweep, WOQ_PERSIST_HINTS=1 — cost a round trip more than
//     they saved in re-reads: 47.9 vs 42.7 us per layer);
//   * attention runs on four consumer waves of the first `heads` workgroups (woq_attn_decode.h) between the qkv and
//     the o projection, on {tag, fp32} granules of q | k | v exactly as in the fused launch (woq_gemv_attn.hip).
// No wave of the kernel reaches an s_barrier after the first instruction: meeting points are LDS counters, the loader
// never waits for anything but ring space. Every wait is bounded (~20 ms, then a sticky status word and no further
// waiting), so a missing producer costs a wrong token, not a hung GPU.
// MEASURED (MI355X, Llama-2-7B shape, profiles/r03ad_persist_*.txt, tools/persist_stamps.py): correct — greedy tokens
// equal to the separate launches', logits within 3e-7 of the largest, repeatable bit for bit, status clear — and
// SLOWER: 40-41 us per layer against 35 (1340 vs 1190 us per token). The stamps say why. The stream is no longer
// the limit: the loader sits in ring-full waits for half of every layer. The layer is a chain of five all-to-all
// hand-offs (qkv -> attention -> o -> gate/up -> down -> qkv'), and each costs 4-6 us on the critical path however the
// weights got there: the slowest workgroup's last tile, a meeting of its consumers (0.5-1 us), the epilogue (0.5-1.3),
// one memory round trip out and one back under a loaded fabric (1.2-2.8 us until every granule of the vector is seen;
// 4 us behind the 2-vs-3-pairs imbalance of gate/up), another meeting (0.3-0.6) — during which no tile of the next
// projection can be touched — and then 1-6 us of tile passes that are bound by SIMD issue (~35 VALU + 2 MFMA per KiB
// tile, three waves per SIMD), not by LDS or HBM. A kernel boundary does the same all-to-all in ~2 us of hardware.
// What it settles for the launches: the decode step's wall is its dependency chain (5 hand-offs x 32 layers) plus the
// per-tile instruction stream, not the weight stream's ramp. Default OFF (WOQ_ENGINE_PERSIST=1 /
// woq_engine_set_persist); kept as a tested, instrumented path.
// Scope: one GPU, int4 blobs without padding, multi-head attention with head_dim 128, no sliding window, one attention
// slice (short contexts), everything resident in 160 KiB of LDS (activation vector of the widest projection + scales
// of the workgroup's strips + ring >= 64 KiB). Everything else keeps the launches (woq_engine.hip).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "woq_attn_decode.h"
#include "woq_gemv_common.h"
#include "woq_gemv_xqs.h"
#include "woq_launch.h"
#include "woq_persist.h"
#include "woq_xq.h"

namespace woq {

#ifndef WOQ_PS_NC
#define WOQ_PS_NC 11
#endif
constexpr int PS_NC = WOQ_PS_NC;              // consumer waves (11: three waves per SIMD, 168 VGPRs each — the attention
                                              // body does not fit 128 without spilling into the epilogues' way)
constexpr int PS_THREADS = 64 * (PS_NC + 1);  // loader + consumers
constexpr int PS_D = 32;          // 1-KiB LDS-DMA pieces in flight (vmcnt is a 6-bit counter)
constexpr int PS_SPIN_LIMIT = 1 << 18;  // LDS polls of ~200 clocks: ~20 ms

// ---- device-side description of the token's work ----
struct PsGemv {  // one projection
  const uint8_t* q;    // [strip][tiles_k][1 KiB]
  const uint8_t* sc;   // [strip][sc_strip bytes]
  const uint8_t* zp;   // [strip][zp_strip bytes] or null
  const float* norm_next;  // weight multiplied into the output before it leaves as the next vector (null = 1)
  int tiles_k, units, n_groups, tpg_shift;
  int sc_strip, zp_strip, pad0, pad1;
};
struct PsLayer {
  PsGemv g[4];  // qkv | o | gate/up (strip pairs) | down
  void* kc;
  void* vc;
};
struct XgVec {  // an activation vector as tagged granules
  unsigned long long* limb;  // [nb][8]: tag16 << 48 | values 2 p, 2 p + 1 of the block: s0 s1 s2 | s0 s1 s2
  unsigned long long* meta;  // [nb]: tag16 << 48 | (e + 128) << 32 | fp32 sum of squares of the raw values
};
struct PsArgs {
  const PsLayer* layers;
  int n_layers, hidden, heads, kv_heads;
  int flags;  // bit 0 bf16 scales, bit 1 scale mode 1, bit 2 zero points, bit 3 fp32 scales
  float eps;
  int spw;
  // LDS layout (bytes)
  int o_zero, o_resid, o_vec, o_u, o_sx, o_ssq, o_sc[4], o_zp[4], o_slab, o_ring;
  int ring_tiles, slab_strips;
  // step state
  const unsigned int* seq;
  const int32_t* pos;
  int* status;
  const float* cs;
  const float* sn;
  // vectors
  XqPtrs x0;          // layer 0's input, written by the embedding launch (plain XQ)
  const float* ssq0;  // its per-block sums of squares
  XgVec xg_hidden, xg_attn, xg_act;
  unsigned long long* qkv_g;  // {tag32, fp32} granules of q | k | v
  unsigned int* hint;         // [4][G] "my outputs are on their way": attention (2 per head) | o | gate/up | down
  float* hidden_buf;          // fp32 residual stream (in: embedding row, out: the last layer's output)
  unsigned long long* stamps;  // diagnostics (null = off): [workgroup][layer * 4 + projection][32] wall-clock stamps
};

// stamp slots. Consumer wave 0: 0 projection begins, 1 hints seen, 2 input vector staged, 3 all consumers staged,
// 4 own tiles done, 5 all consumers' tiles done, 6 epilogue stores issued, 7 attention done (qkv only); 16 / 17: clocks
// of its tile passes spent waiting for tiles to land / fetching and computing.
// Loader: 8 first tile about to be issued, 9 last tile issued, 10 waits for ring space / scale area, 11 clocks in them.
#define PS_STAMP(slot)                                                                            \
  do {                                                                                            \
    if (a.stamps != nullptr && lane == 0)                                                         \
      a.stamps[((size_t)b * (a.n_layers * 4) + (l * 4 + k)) * 32 + (slot)] = wall_clock64();      \
  } while (0)

typedef __attribute__((address_space(3))) volatile int lds_vint;
// The work description is read through the CONSTANT address space: written by the host before the first launch and never
// again, so its (uniform) loads are scalar loads. As plain global loads they would be vector loads followed by
// s_waitcnt vmcnt(0) — in the loader that drains the whole LDS-DMA queue at every projection.
typedef const __attribute__((address_space(4))) PsLayer* ps_layers_t;
typedef const __attribute__((address_space(4))) PsGemv& ps_gemv_t;
// ctl words: [0] landed tiles, [1] consumer meeting count, [3] hint epoch, [4] attention meeting count, [5] gave up,
//            [8 + c] tiles consumer c is done with (everything below this global tile index)

__device__ __forceinline__ void ps_dma_1k(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void ps_dma_4k(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void ps_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void ps_span(int b, int G, int units, int& u0, int& nu) {
  u0 = (int)(((long long)b * units) / G);
  nu = (int)(((long long)(b + 1) * units) / G) - u0;
}
__device__ __forceinline__ int row16_min_i32(int v) {
  v = min(v, WOQ_DPP_I32(v, 0xB1));
  v = min(v, WOQ_DPP_I32(v, 0x4E));
  v = min(v, WOQ_DPP_I32(v, 0x141));
  return min(v, WOQ_DPP_I32(v, 0x140));
}
__device__ __forceinline__ unsigned int ps_tag(unsigned int seq, int layer, int kind) {
  return (seq << 10) | ((unsigned int)layer << 3) | (unsigned int)kind;  // kinds: 0 qkv 1 attention 2 o 3 gate/up 4 down
}

// 16 lanes of one DPP row, lane j holding value 16 blk + j: block `blk` as nine granules (woq_xq.h's conversion)
__device__ __forceinline__ void xg_emit16(float y, const XgVec& o, int blk, int j, unsigned int tag16, float ss) {
  const float amax = row16_max(fabsf(y));
  int e = 0;
  if (amax > 0.f && amax < INFINITY) e = max(-100, min(100, __builtin_amdgcn_frexp_expf(amax)));
  const uint32_t Q = __float_as_uint(fmaf(y, ldexpf(1.f, 21 - e), 12582912.f));
  const int v = (int)(Q & 0x7fffffu) - (1 << 22);
  const int s0 = (int)((uint32_t)v << 24) >> 24;
  const int v1 = (v - s0) >> 8;
  const int s1 = (int)((uint32_t)v1 << 24) >> 24;
  const int s2 = (v1 - s1) >> 8;
  const unsigned int mine = (unsigned int)(s0 & 255) | ((unsigned int)(s1 & 255) << 8) | ((unsigned int)(s2 & 255) << 16);
  const unsigned int next = (unsigned int)WOQ_DPP_I32((int)mine, 0xB1);  // lane j ^ 1
  if (!(j & 1))
    __hip_atomic_store(o.limb + (size_t)blk * 8 + (j >> 1),
                       (unsigned long long)mine | ((unsigned long long)next << 24) | ((unsigned long long)tag16 << 48),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (j == 0)
    __hip_atomic_store(o.meta + blk,
                       (unsigned long long)__float_as_uint(ss) | ((unsigned long long)((e + 128) & 255) << 32) |
                           ((unsigned long long)tag16 << 48),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ================================================ loader ================================================
__device__ __forceinline__ void ps_loader(const PsArgs& a, unsigned char* smem, lds_vint* ctl, int lane, int G, int b) {
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const int R = a.ring_tiles;
  int P = 0, j = 0, slot = 0, m_seen = 0, pub = 0;
  int P0c = 0, j0c = 0, Tc = 0, P0p = 0, j0p = 0, Tp = 0;  // current / previous projection: first tile's piece, index
  int end_k[4] = {0, 0, 0, 0};  // end tile index of the last projection of each kind (the tenant of its scale area)
  bool ok = true;
  unsigned long long stall_n = 0, stall_t = 0;
  int cur_op = 0;
  auto publish = [&](int n) {
    if (n > pub) {
      pub = n;
      if (lane == 0) ctl[0] = n;
    }
  };
  auto landed_to = [&](int lp) {  // pieces below lp have landed (vmcnt retires in order)
    if (lp > P0c)
      publish(j0c + (lp - P0c));
    else if (lp > P0p)
      publish(j0p + min(lp - P0p, Tp));
  };
  auto min_done = [&]() {
    int v = lane < PS_NC ? ctl[8 + lane] : 0x7fffffff;
    return uni(row16_min_i32(v));
  };
  auto wait_consumed = [&](int need) {  // every tile below `need` has been consumed
    if (m_seen >= need) return;
    m_seen = min_done();
    if (m_seen >= need || !ok) return;
    ps_wait_vmcnt<0>();  // stalled anyway: whatever is in flight may land and be seen
    landed_to(P);
    const unsigned long long ts = a.stamps != nullptr ? wall_clock64() : 0ull;
    for (int spins = 0;; ++spins) {
      m_seen = min_done();
      if (m_seen >= need) {
        if (a.stamps != nullptr) {
          const unsigned long long te = wall_clock64();
          if (stall_n < 2 && lane == 0) {  // slots 12..15: begin / end of the projection's first two waits
            unsigned long long* sp = a.stamps + ((size_t)b * (a.n_layers * 4) + cur_op) * 32 + 12 + 2 * stall_n;
            sp[0] = ts, sp[1] = te;
          }
          ++stall_n, stall_t += te - ts;
        }
        break;
      }
      if (ctl[5] != 0 || spins > PS_SPIN_LIMIT) {
        ok = false;
        if (lane == 0) {
          ctl[5] = 1;
          atomicOr(a.status, 1);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  };
  auto dma_range = [&](const uint8_t* gsrc, int bytes, uint32_t dst) {  // a small contiguous array, 1-KiB pieces
    for (int off = 0; off < bytes; off += 1024) {
      ps_dma_1k(gsrc + min(off + lane * 16, bytes - 16), uni(dst + (uint32_t)off));
      ++P;
      ps_wait_vmcnt<PS_D - 1>();
    }
  };
  for (int l = 0; l < a.n_layers; ++l) {
    for (int k = 0; k < 4; ++k) {
      ps_gemv_t g = ((ps_layers_t)(unsigned long long)a.layers)[l].g[k];
      const int cb = k == 2 ? 2 : 1;
      int u0, nu;
      ps_span(b, G, g.units, u0, nu);
      const int T = nu * cb * g.tiles_k;
      if (T == 0) continue;
      cur_op = l * 4 + k;
      {  // projections start at multiples of four tiles: a burst never straddles the ring's end
        const int pad = (-j) & 3;
        j += pad;
        slot += pad;
        if (slot >= R) slot -= R;
      }
      wait_consumed(end_k[k]);  // the scale area's previous tenant (this projection, one layer up) is finished
      dma_range(g.sc + (size_t)u0 * cb * g.sc_strip, nu * cb * g.sc_strip, lds0 + (uint32_t)a.o_sc[k]);
      if (g.zp != nullptr) dma_range(g.zp + (size_t)u0 * cb * g.zp_strip, nu * cb * g.zp_strip, lds0 + (uint32_t)a.o_zp[k]);
      P0p = P0c, j0p = j0c, Tp = Tc;
      P0c = P, j0c = j, Tc = T;
      PS_STAMP(8);
      const uint8_t* src = g.q + (size_t)u0 * cb * g.tiles_k * 1024 + lane * 16;
      int t = 0;
      for (; t + 4 <= T; t += 4) {
        wait_consumed(j + 4 - R);
        ps_dma_4k(src, uni(lds0 + (uint32_t)a.o_ring + (uint32_t)slot * 1024u));
        src += 4096;
        slot += 4;
        if (slot >= R) slot = 0;
        j += 4;
        P += 4;
        ps_wait_vmcnt<PS_D - 4>();
        landed_to(P - (PS_D - 4));
      }
      for (; t < T; ++t) {
        wait_consumed(j + 1 - R);
        ps_dma_1k(src, uni(lds0 + (uint32_t)a.o_ring + (uint32_t)slot * 1024u));
        src += 1024;
        if (++slot == R) slot = 0;
        ++j;
        ++P;
        ps_wait_vmcnt<PS_D - 1>();
        landed_to(P - (PS_D - 1));
      }
      end_k[k] = j;
      PS_STAMP(9);
      if (a.stamps != nullptr && lane == 0) {
        unsigned long long* sp = a.stamps + ((size_t)b * (a.n_layers * 4) + (l * 4 + k)) * 32;
        sp[10] = stall_n, sp[11] = stall_t;
      }
      stall_n = stall_t = 0;
    }
  }
  ps_wait_vmcnt<0>();
  landed_to(P);
}

// ================================================ consumers ================================================
struct PsWaves {  // a consumer wave's view of the workgroup's meeting points
  lds_vint* ctl;
  int lane;
  int* status;
  mutable int bar, abar;
  mutable bool ok;
  __device__ __forceinline__ void give_up() const {
    ok = false;
    if (lane == 0) {
      ctl[5] = 1;
      atomicOr(status, 1);
    }
  }
  // all PS_NC consumers (word 1) / the four attention waves (word 4): arrive, then wait for everybody
  __device__ __forceinline__ void meet(int word, int n, int& epoch) const {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)&ctl[word], 1, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
    ++epoch;
    // (polling without the sleep was measured slower: the pollers take issue slots and LDS cycles from the waves
    // that are still working — 415 vs 381 us per 8-layer step)
    for (int spins = 0; ok && ctl[word] < n * epoch; ++spins) {
      if (ctl[5] != 0 || spins > PS_SPIN_LIMIT) {
        give_up();
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  }
  __device__ __forceinline__ void meet_all() const { meet(1, PS_NC, bar); }
};

struct PsAttnEnv {  // woq_attn_decode.h's environment: consumer waves 0..3 of an attention workgroup
  const PsWaves& w;
  int tid_;
  XgVec out;
  unsigned int tag16;
  __device__ __forceinline__ int tid() const { return tid_; }
  __device__ __forceinline__ void sync() const { w.meet(4, 4, w.abar); }
  __device__ __forceinline__ void put(float v, int idx) const { xg_emit16(v, out, idx >> 4, idx & 15, tag16, 0.f); }
};

// the input vector of a projection -> LDS: limbs [nb][3][16], block factors u [nb], (block sums sx [nb]), (sums of
// squares ssq [nb]). Granule form: every lane re-reads the granules whose tag is not yet `tag16`.
__device__ __forceinline__ void ps_stage_xg(const XgVec& x, int nb, unsigned int tag16, unsigned char* limbs, float* lu,
                                            float* lsx, float* lssq, bool asym, int c, const PsWaves& w) {
  const int lane = w.lane;
  const int total = nb * 8;
  const int per_wave = (((total + PS_NC - 1) / PS_NC) + 63) & ~63;
  const int g_begin = c * per_wave, g_end = min(total, g_begin + per_wave);
  unsigned short* l16 = (unsigned short*)limbs;
  // this lane's first meta granule travels with the first limb granules (one round trip, not two)
  unsigned long long m_first = 0;
  if (c * 64 + lane < nb) m_first = __hip_atomic_load(x.meta + c * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int base = g_begin; base < g_end; base += 512) {
    unsigned long long gv[8];
    unsigned int bad = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int idx = base + 64 * r + lane;
      gv[r] = 0;
      if (idx < g_end) {
        gv[r] = __hip_atomic_load(x.limb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned int)(gv[r] >> 48) != tag16) bad |= 1u << r;
      }
    }
    if (__any(bad != 0)) {
      const unsigned long long t0 = wall_clock64();
      while (w.ok) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if ((bad >> r) & 1u) {
            gv[r] = __hip_atomic_load(x.limb + base + 64 * r + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned int)(gv[r] >> 48) == tag16) bad &= ~(1u << r);
          }
        if (!__any(bad != 0)) break;
        if (w.ctl[5] != 0 || wall_clock64() - t0 > 2000000ull) {
          w.give_up();
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int idx = base + 64 * r + lane;
      if (idx < g_end) {  // whole 8-lane groups (one block each): g_end is a multiple of 8
        const unsigned int lo = (unsigned int)gv[r], hi = (unsigned int)(gv[r] >> 24);
        const int blk = idx >> 3, p = idx & 7;
        unsigned short* d = l16 + (size_t)blk * 24 + p;
        d[0] = (unsigned short)((lo & 0xffu) | ((hi & 0xffu) << 8));
        d[8] = (unsigned short)(((lo >> 8) & 0xffu) | (((hi >> 8) & 0xffu) << 8));
        d[16] = (unsigned short)(((lo >> 16) & 0xffu) | (((hi >> 16) & 0xffu) << 8));
        if (asym) {  // the block's sum of fixed-point values, exact in int32
          const int va = ((int)(lo << 24) >> 24) + (((int)(lo << 16) >> 24) << 8) + (((int)(lo << 8) >> 24) << 16);
          const int vb = ((int)(hi << 24) >> 24) + (((int)(hi << 16) >> 24) << 8) + (((int)(hi << 8) >> 24) << 16);
          int s = va + vb;
          s += WOQ_DPP_I32(s, 0xB1);
          s += WOQ_DPP_I32(s, 0x4E);
          s += WOQ_DPP_I32(s, 0x141);
          if (p == 0) lsx[blk] = (float)s;
        }
      }
    }
  }
  for (int b0 = c * 64; b0 < nb; b0 += PS_NC * 64) {
    const int blk = b0 + lane;
    unsigned long long m = 0;
    bool bad = false;
    if (blk < nb) {
      m = b0 == c * 64 ? m_first : __hip_atomic_load(x.meta + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bad = (unsigned int)(m >> 48) != tag16;
    }
    if (__any(bad)) {
      const unsigned long long t0 = wall_clock64();
      while (w.ok) {
        if (bad) {
          m = __hip_atomic_load(x.meta + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad = (unsigned int)(m >> 48) != tag16;
        }
        if (!__any(bad)) break;
        if (w.ctl[5] != 0 || wall_clock64() - t0 > 2000000ull) {
          w.give_up();
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (blk < nb) {
      const int e8 = (int)((m >> 32) & 0xffu);  // e + 128: u = 2^(e - 25)
      lu[blk] = __uint_as_float((unsigned int)(e8 - 128 - 25 + 127) << 23);
      if (lssq != nullptr) lssq[blk] = __uint_as_float((unsigned int)m);
    }
  }
}

__device__ __forceinline__ void ps_stage_plain(const XqPtrs& x, const float* ssq, int nb, unsigned char* limbs, float* lu,
                                               float* lsx, float* lssq, bool asym, int c, int lane) {
  const int bytes = nb * 48;
  for (int off = (c * 64 + lane) * 16; off < bytes; off += PS_NC * 1024)
    *(u32x4*)(limbs + off) = *(const u32x4*)(x.limbs + off);
  for (int i = c * 64 + lane; i < nb; i += PS_NC * 64) {
    lu[i] = x.u[i];
    if (asym) lsx[i] = x.sx[i];
    if (lssq != nullptr) lssq[i] = ssq[i];
  }
}

template <typename KV, int SMODE, bool ASYM>
__device__ __forceinline__ void ps_consumer(const PsArgs& a, unsigned char* smem, lds_vint* ctl, int lane, int c, int G,
                                            int b) {
  const PsWaves w{ctl, lane, a.status, 0, 0, true};
  const int R = a.ring_tiles;
  const bool bf = (a.flags & 1) != 0, s32 = (a.flags & 8) != 0;
  constexpr bool asym = ASYM;
  unsigned char* zero_blk = smem + a.o_zero;
  float* resid = (float*)(smem + a.o_resid);
  unsigned char* limbs = smem + a.o_vec;
  float* lu = (float*)(smem + a.o_u);
  float* lsx = (float*)(smem + a.o_sx);
  float* lssq = (float*)(smem + a.o_ssq);
  const unsigned char* ring = smem + a.o_ring;
  const int S = a.slab_strips;
  const unsigned int seq = a.seq[0];
  const int i16 = lane & 15, kq = lane >> 4;
  const bool a_live = (i16 >> 2) == kq && (i16 & 3) != 3;
  const unsigned char* a_base = a_live ? limbs + kq * 48 + (i16 & 3) * 16 : zero_blk + kq * 16;
  const int a_step_t = a_live ? 384 : 0, a_step_h = a_live ? 192 : 0;
  const i32x4 izero = {0, 0, 0, 0};
  int j0 = 0, slot0 = 0, landed_seen = 0, hint_epoch = 0, op_no = 0;
  // diagnostics (consumer wave 0): stamps stay in registers and are stored once per projection — a store per stamp
  // would wait for the previous stamp's store and smear every phase by a memory round trip
  const bool stamping = a.stamps != nullptr && c == 0;
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_wait = 0, t_pass = 0;
#define PS_T(n)                          \
  do {                                   \
    if (stamping) ts[n] = wall_clock64(); \
  } while (0)

  for (int l = 0; l < a.n_layers; ++l) {
    for (int k = 0; k < 4; ++k, ++op_no) {
      ps_gemv_t g = ((ps_layers_t)(unsigned long long)a.layers)[l].g[k];
      const int cb = k == 2 ? 2 : 1;
      int u0, nu;
      ps_span(b, G, g.units, u0, nu);
      const int tiles_k = g.tiles_k;
      const int T = nu * cb * tiles_k;
      const unsigned int tag_out = ps_tag(seq, l, k == 0 ? 0 : k + 1);
      if (T > 0) {
        {
          const int pad = (-j0) & 3;
          j0 += pad;
          slot0 += pad;
          if (slot0 >= R) slot0 -= R;
        }
        const int nb = tiles_k * 8;
        const bool norm = k == 0 || k == 2;
        PS_T(0);
        // ---- 1. the input vector ----
        if (l == 0 && k == 0) {
          ps_stage_plain(a.x0, a.ssq0, nb, limbs, lu, lsx, lssq, asym, c, lane);
        } else {
          const XgVec xin = k == 1 ? a.xg_attn : (k == 3 ? a.xg_act : a.xg_hidden);
          const unsigned int tag_in = k == 0 ? ps_tag(seq, l - 1, 4) : ps_tag(seq, l, k);
          const unsigned int* hint = a.hint + (size_t)(k == 0 ? 3 : k - 1) * G;
          const int n_hint = k == 1 ? min(2 * a.heads, G) : G;
          ++hint_epoch;
          if ((a.flags & 16) != 0) {
            // no hint phase: the sweep below polls the granules themselves (one round trip less, more re-reads)
          } else if (c == 0) {  // one wave polls the producers' hint words for the whole workgroup
            const unsigned long long t0 = wall_clock64();
            while (w.ok) {
              bool good = true;
              for (int i = lane; i < n_hint; i += 64)
                good = good && __hip_atomic_load(hint + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag_in;
              if (__all(good)) break;
              if (ctl[5] != 0 || wall_clock64() - t0 > 2000000ull) {
                w.give_up();
                break;
              }
              __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0) ctl[3] = hint_epoch;
            PS_T(1);
          } else {
            for (int spins = 0; w.ok && ctl[3] < hint_epoch; ++spins) {
              if (ctl[5] != 0 || spins > PS_SPIN_LIMIT) {
                w.give_up();
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          ps_stage_xg(xin, nb, tag_in & 0xffffu, limbs, lu, lsx, norm ? lssq : nullptr, asym, c, w);
        }
        PS_T(2);
        const int first = min(c, T);
        if (lane == 0) ctl[8 + c] = j0 + first;
        float* slab = (float*)(smem + a.o_slab) + (size_t)(op_no & 1) * PS_NC * S * 16;
        for (int i = lane; i < nu * cb * 16; i += 64) slab[((size_t)c * S) * 16 + i] = 0.f;
        // what the epilogue lanes need from memory: asked for now, used after the tiles
        float gw[2] = {1.f, 1.f};
        if (c == 0 && g.norm_next != nullptr) {
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int ul = ps * 4 + (lane >> 4);
            if (ul < nu) gw[ps] = g.norm_next[(u0 + ul) * 16 + i16];
          }
        }
        w.meet_all();
        PS_T(3);
        float inv = 1.f;
        if (norm && c == 0) {
          float s = 0.f;
          for (int i = lane; i < nb; i += 64) s += lssq[i];
          s = wave_sum_dpp(s);
          inv = 1.0f / sqrtf(s / (float)(tiles_k * 128) + a.eps);  // HF LlamaRMSNorm
        }
        // ---- 2. the tiles, round-robin, two per pass: every LDS read of both is in flight before the first MFMA ----
        const unsigned char* lsc = smem + a.o_sc[k];
        const unsigned char* lzp = smem + a.o_zp[k];
        const int sc_strip = g.sc_strip, zp_strip = g.zp_strip, tpg_shift = g.tpg_shift, n_groups = g.n_groups;
        int sidx = 0, kt = c;
        while (kt >= tiles_k) {
          kt -= tiles_k;
          ++sidx;
        }
        int slot = slot0 + c;
        while (slot >= R) slot -= R;
        float acc = 0.f;
        int acc_sidx = sidx;
        auto flush = [&]() {  // the four lane quarters hold the four blocks' shares of each column
          float v = acc;
          v += xqs_swap32(v);
          v += xqs_swap16(v);
          if (lane < 16) slab[((size_t)c * S + acc_sidx) * 16 + lane] = v;
          acc = 0.f;
        };
        struct Tile {
          u32x4 wv;
          i32x4 a0, a1;
          float u0, u1, s0, s1;
          uint32_t r0, r1, z0, z1;  // raw scale words / zero points
          int sidx;
        };
        const bool xk_noa = (a.flags & 64) != 0, xk_nomath = (a.flags & 32) != 0, xk_now = (a.flags & 128) != 0;
        auto fetch = [&](Tile& t, int kt_, int sidx_, int slot_) {
          t.sidx = sidx_;
          t.wv = u32x4{1u, 2u, 3u, 4u};
          if (!xk_now) t.wv = *(const u32x4*)(ring + (size_t)slot_ * 1024 + lane * 16);
          t.a0 = t.a1 = izero;
          if (!xk_noa) {
            t.a0 = *(const i32x4*)(a_base + kt_ * a_step_t);
            t.a1 = *(const i32x4*)(a_base + kt_ * a_step_t + a_step_h);
          }
          t.u0 = lu[kt_ * 8 + kq], t.u1 = lu[kt_ * 8 + 4 + kq];
          if constexpr (ASYM) t.s0 = lsx[kt_ * 8 + kq], t.s1 = lsx[kt_ * 8 + 4 + kq];
          const unsigned char* scp = lsc + (size_t)sidx_ * sc_strip;
          const unsigned char* zpp = lzp + (size_t)sidx_ * zp_strip;
          if constexpr (SMODE == 0) {
            const int gi = min(kt_ >> tpg_shift, n_groups - 1);
            t.r0 = s32 ? *(const uint32_t*)(scp + (gi * 16 + i16) * 4) : *(const uint16_t*)(scp + (gi * 16 + i16) * 2);
            if constexpr (ASYM) t.z0 = zpp[gi * 16 + i16];
          } else {  // this lane quarter's 32-k group of each half
            const int o = (kt_ * 16 + i16) * 4 + (kq >> 1);
            if (s32) {
              t.r0 = *(const uint32_t*)(scp + o * 4), t.r1 = *(const uint32_t*)(scp + (o + 2) * 4);
            } else {
              t.r0 = *(const uint16_t*)(scp + o * 2), t.r1 = *(const uint16_t*)(scp + (o + 2) * 2);
            }
            if constexpr (ASYM) t.z0 = zpp[o], t.z1 = zpp[o + 2];
          }
        };
        auto mac = [&](const Tile& t) {
          if (t.sidx != acc_sidx) {
            flush();
            acc_sidx = t.sidx;
          }
          const u32x4 wv = t.wv;
          if (xk_nomath) {
            acc += __uint_as_float((wv.x ^ wv.y) ^ (wv.z ^ wv.w)) + __int_as_float(t.a0.x ^ t.a1.y) + t.u0 + __uint_as_float(t.r0);
            return;
          }
          const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u), (int)((wv.y << 4) & 0xf0f0f0f0u),
                            (int)(wv.y & 0xf0f0f0f0u)};
          const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u), (int)((wv.w << 4) & 0xf0f0f0f0u),
                            (int)(wv.w & 0xf0f0f0f0u)};
          const i32x4 d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(t.a0, b0, izero, 0, 0, 0);
          const i32x4 d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(t.a1, b1, izero, 0, 0, 0);
          float f0 = digit_combine(d0), f1 = digit_combine(d1);
          if constexpr (SMODE == 0) {
            if constexpr (ASYM) {  // the weights carry 16 * q: the zero point enters as 16 * zp
              const float z16 = -16.f * (float)((int)t.z0 - 8);
              f0 = fmaf(z16, t.s0, f0);
              f1 = fmaf(z16, t.s1, f1);
            }
            const float sc = s32 ? __uint_as_float(t.r0) : tscale16(t.r0, bf);
            acc = fmaf(sc, fmaf(f0, t.u0, f1 * t.u1), acc);
          } else {
            if constexpr (ASYM) {
              f0 = fmaf(-16.f * (float)((int)t.z0 - 8), t.s0, f0);
              f1 = fmaf(-16.f * (float)((int)t.z1 - 8), t.s1, f1);
            }
            const float sc0 = s32 ? __uint_as_float(t.r0) : tscale16(t.r0, bf);
            const float sc1 = s32 ? __uint_as_float(t.r1) : tscale16(t.r1, bf);
            acc = fmaf(sc0 * t.u0, f0, fmaf(sc1 * t.u1, f1, acc));
          }
        };
        auto advance = [&]() {
          kt += PS_NC;
          while (kt >= tiles_k) {
            kt -= tiles_k;
            ++sidx;
          }
          slot += PS_NC;
          if (slot >= R) slot -= R;
        };
        for (int i = c; i < T; i += 2 * PS_NC) {
          const bool two = i + PS_NC < T;
          const int j = j0 + (two ? i + PS_NC : i);  // tiles land in order
          const unsigned long long tw0 = stamping ? wall_clock64() : 0ull;
          if (j >= landed_seen) {
            for (int spins = 0; w.ok; ++spins) {
              landed_seen = ctl[0];
              if (landed_seen > j) break;
              if (ctl[5] != 0 || spins > PS_SPIN_LIMIT) {
                w.give_up();
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          const unsigned long long tp0 = stamping ? wall_clock64() : 0ull;
          if (stamping) t_wait += tp0 - tw0;
          Tile t0, t1;
          fetch(t0, kt, sidx, slot);
          advance();
          if (two) {
            fetch(t1, kt, sidx, slot);
            advance();
          }
          asm volatile("" ::: "memory");
          if (lane == 0) ctl[8 + c] = j0 + min(i + 2 * PS_NC, T);  // the tiles' bytes are on their way to registers
          mac(t0);
          if (two) mac(t1);
          if (stamping) t_pass += wall_clock64() - tp0;
        }
        // everything below the next projection's first tile is consumed as far as this wave is concerned (the index
        // space is padded to a multiple of four there): the loader may fill the ring before anybody starts on it
        if (lane == 0) ctl[8 + c] = (j0 + T + 3) & ~3;
        if (c < T) flush();
        PS_T(4);
        w.meet_all();
        PS_T(5);
        // ---- 3. the workgroup's column units: 16 lanes each, four at a time ----
        if (c == 0) {
          for (int ps = 0; ps * 4 < nu; ++ps) {
            const int ul = ps * 4 + (lane >> 4);
            if (ul < nu) {
              float v = 0.f, up = 0.f;
#pragma unroll
              for (int c2 = 0; c2 < PS_NC; ++c2) {
                v += slab[((size_t)c2 * S + ul * cb) * 16 + i16];
                if (cb == 2) up += slab[((size_t)c2 * S + ul * cb + 1) * 16 + i16];
              }
              const int unit = u0 + ul, n = unit * 16 + i16;
              if (k == 0) {
                __hip_atomic_store(a.qkv_g + n, ((unsigned long long)tag_out << 32) | __float_as_uint(v * inv),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              } else if (k == 2) {
                v *= inv;
                up *= inv;
                xg_emit16(v / (1.0f + __expf(-v)) * up, a.xg_act, unit, i16, tag_out & 0xffffu, 0.f);
              } else {
                const float r = resid[ul * 16 + i16] + v;
                resid[ul * 16 + i16] = r;
                a.hidden_buf[n] = r;
                if (!(k == 3 && l + 1 == a.n_layers)) {
                  const float ss = row16_sum(r * r);
                  const float gn = ps < 2 ? gw[ps] : (g.norm_next != nullptr ? g.norm_next[n] : 1.f);
                  xg_emit16(r * gn, a.xg_hidden, unit, i16, tag_out & 0xffffu, ss);
                }
              }
            }
          }
        }
        PS_T(6);
        j0 += T;
        slot0 += T % R;
        if (slot0 >= R) slot0 -= R;
      }
      if (k != 0 && c == 0 && lane == 0)
        __hip_atomic_store(a.hint + (size_t)k * G + b, tag_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == 0 && b < a.heads) {
        // ---- attention of head b on consumer waves 0..3; the others wait: the scratch is the vector area ----
        if (c < 4) {
          const unsigned int tag_attn = ps_tag(seq, l, 1);
          const PsAttnEnv env{w, c * 64 + lane, a.xg_attn, tag_attn & 0xffffu};
          attn_decode_env<KV, 128, false>((float*)limbs, b, 0, 1, AttnGranule{a.qkv_g, tag_out, a.status},
                                          (KV*)((ps_layers_t)(unsigned long long)a.layers)[l].kc,
                                          (KV*)((ps_layers_t)(unsigned long long)a.layers)[l].vc, a.pos, a.cs, a.sn, a.heads,
                                          a.kv_heads, 0, a.spw, (float*)nullptr, env);
          if (c < 2 && lane == 0)  // waves 0 and 1 hold the 128 output values: one hint each
            __hip_atomic_store(a.hint + 2 * b + c, tag_attn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        w.meet_all();
        PS_T(7);
      }
      if (stamping && lane == 0) {
        unsigned long long* sp = a.stamps + ((size_t)b * (a.n_layers * 4) + (l * 4 + k)) * 32;
#pragma unroll
        for (int n = 0; n < 8; ++n) sp[n] = ts[n];
        sp[16] = t_wait, sp[17] = t_pass;
        t_wait = t_pass = 0;
      }
    }
  }
}

template <typename KV, int SMODE, bool ASYM>
__global__ __launch_bounds__(PS_THREADS) void persist_kernel(const PsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = uni(tid >> 6);
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  lds_vint* ctl = (lds_vint*)(__attribute__((address_space(3))) void*)smem;
  if (tid < 32) ctl[tid] = 0;
  if (tid < 16) ((uint32_t*)(smem + a.o_zero))[tid] = 0u;
  {  // this workgroup's elements of the residual stream stay in LDS for the whole token
    int uh0, nuh;
    ps_span(b, G, a.hidden >> 4, uh0, nuh);
    float* resid = (float*)(smem + a.o_resid);
    for (int i = tid; i < nuh * 16; i += PS_THREADS) resid[i] = a.hidden_buf[uh0 * 16 + i];
  }
  __syncthreads();  // the launch's only s_barrier: from here on the waves have different jobs
  if (wid == 0)
    ps_loader(a, smem, ctl, lane, G, b);
  else
    ps_consumer<KV, SMODE, ASYM>(a, smem, ctl, lane, wid - 1, G, b);
}

// ================================================ host ================================================
struct Persist {
  PsArgs args;
  PsLayer* d_layers = nullptr;
  void* d_vecs = nullptr;
  int grid = 0, kv_dtype = 0;
  size_t lds = 0;
};

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static bool blob_ok(const woq_blob_header& h, const woq_blob_header& ref) {
  if (h.weight_type != WOQ_W_INT4_CLIP || h.off_shuffle != 0 || h.K != h.Kpad || h.N != h.Npad) return false;
  if ((h.K % WOQ_TILE_K) != 0 || (h.N % 16) != 0) return false;
  if (h.scale_mode != ref.scale_mode || h.scale_type != ref.scale_type || (h.off_zp != 0) != (ref.off_zp != 0)) return false;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return false;
  }
  return true;
}

static void fill_gemv(PsGemv& g, const void* blob, const woq_blob_header& h, int cb, const float* norm_next) {
  const uint8_t* b = (const uint8_t*)blob;
  const int esz = h.scale_type == WOQ_F32 ? 4 : 2;
  g.q = b + h.off_q;
  g.sc = b + h.off_scale;
  g.zp = h.off_zp ? b + h.off_zp : nullptr;
  g.norm_next = norm_next;
  g.tiles_k = h.Kpad / WOQ_TILE_K;
  g.units = h.Npad / 16 / cb;
  g.n_groups = h.n_groups;
  g.tpg_shift = 0;
  if (h.scale_mode == 0 && h.n_groups > 1)
    for (int tpg = h.group / WOQ_TILE_K; tpg > 1; tpg >>= 1) ++g.tpg_shift;
  const int per_strip = h.scale_mode == 0 ? h.n_groups * 16 : g.tiles_k * 64;
  g.sc_strip = per_strip * esz;
  g.zp_strip = h.off_zp ? per_strip : 0;
  g.pad0 = g.pad1 = 0;
}

void persist_destroy(Persist* p) {
  if (!p) return;
  if (p->d_layers) hipFree(p->d_layers);
  if (p->d_vecs) hipFree(p->d_vecs);
  delete p;
}

// nullptr (and *why set) when the model / device is outside the kernel's scope
Persist* persist_create(const PersistDesc& d, std::string* why) {
  auto no = [&](const char* m) -> Persist* {
    if (why) *why = m;
    return nullptr;
  };
  if (d.layers < 1 || d.layers >= 128) return no("layer count");
  if (d.head_dim != 128 || d.window != 0 || d.attn_splits > 1 || d.heads != d.kv_heads)
    return no("attention shape (multi-head, head_dim 128, one slice, no window)");
  if (d.kv_dtype != WOQ_F16 && d.kv_dtype != WOQ_BF16 && d.kv_dtype != WOQ_FP8_E4M3) return no("kv dtype");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    return no("device query");
  const int G = cus;
  if (d.heads > G || 2 * d.heads > G) return no("more heads than compute units");
  const woq_blob_header& ref = d.lw[0].qkv_hdr;
  for (int l = 0; l < d.layers; ++l) {
    const woq_layer_weights& w = d.lw[l];
    if (!blob_ok(w.qkv_hdr, ref) || !blob_ok(w.o_hdr, ref) || !blob_ok(w.gate_up_hdr, ref) || !blob_ok(w.down_hdr, ref))
      return no("blob layout");
    if ((w.gate_up_hdr.Npad / 16) & 1) return no("gate/up strips");
    if (w.o_hdr.N != d.hidden || w.down_hdr.N != d.hidden || w.qkv_hdr.K != d.hidden || w.gate_up_hdr.K != d.hidden)
      return no("shapes");
  }
  std::vector<PsLayer> hl(d.layers);
  int max_units[4] = {0, 0, 0, 0}, max_sc[4] = {0, 0, 0, 0}, max_zp[4] = {0, 0, 0, 0}, nb_max = 0;
  for (int l = 0; l < d.layers; ++l) {
    const woq_layer_weights& w = d.lw[l];
    fill_gemv(hl[l].g[0], w.qkv_blob, w.qkv_hdr, 1, nullptr);
    fill_gemv(hl[l].g[1], w.o_blob, w.o_hdr, 1, w.ln2);
    fill_gemv(hl[l].g[2], w.gate_up_blob, w.gate_up_hdr, 2, nullptr);
    fill_gemv(hl[l].g[3], w.down_blob, w.down_hdr, 1, l + 1 < d.layers ? d.lw[l + 1].ln1 : nullptr);
    hl[l].kc = d.kcache + (size_t)l * d.kv_layer_bytes;
    hl[l].vc = d.vcache + (size_t)l * d.kv_layer_bytes;
    for (int k = 0; k < 4; ++k) {
      const PsGemv& g = hl[l].g[k];
      const int cb = k == 2 ? 2 : 1;
      const int mu = (g.units + G - 1) / G;
      max_units[k] = std::max(max_units[k], mu);
      max_sc[k] = std::max(max_sc[k], mu * cb * g.sc_strip);
      max_zp[k] = std::max(max_zp[k], mu * cb * g.zp_strip);
      nb_max = std::max(nb_max, g.tiles_k * 8);
    }
  }
  const bool asym = ref.off_zp != 0;
  Persist* p = new Persist();
  PsArgs& a = p->args;
  a.n_layers = d.layers;
  a.hidden = d.hidden;
  a.heads = d.heads;
  a.kv_heads = d.kv_heads;
  a.flags = (ref.scale_type == WOQ_BF16 ? 1 : 0) | (ref.scale_mode == 1 ? 2 : 0) | (asym ? 4 : 0) |
            (ref.scale_type == WOQ_F32 ? 8 : 0);
  a.eps = d.eps;
  a.spw = attn_dec_spw(d.max_ctx);
  int off = 256;
  a.o_zero = off, off += 64;
  a.o_resid = off, off += ((d.hidden / 16 + G - 1) / G) * 64;
  a.o_vec = off;
  const int vec_bytes = std::max(nb_max * 48, (int)attn_dec_lds_floats(128, d.max_ctx) * 4);
  off += round_up(vec_bytes, 16);
  a.o_u = off, off += nb_max * 4;
  a.o_sx = off, off += asym ? nb_max * 4 : 0;
  a.o_ssq = off, off += (d.hidden / 16) * 4;
  off = round_up(off, 16);
  for (int k = 0; k < 4; ++k) {
    a.o_sc[k] = off, off += round_up(max_sc[k], 1024);
    a.o_zp[k] = off, off += round_up(max_zp[k], 1024);
  }
  a.slab_strips = std::max(std::max(max_units[0], max_units[1]), std::max(2 * max_units[2], max_units[3]));
  a.o_slab = off, off += 2 * PS_NC * a.slab_strips * 64;
  off = round_up(off, 1024);
  a.o_ring = off;
  a.ring_tiles = ((160 * 1024 - off) / 1024) & ~3;
  if (a.ring_tiles < 2 * PS_D) {
    delete p;
    return no("LDS: activation vector + scales leave no room for the weight ring");
  }
  // hint words before the sweep: measured slower than sweeping straight away (47.9 vs 42.7 us per layer): off unless asked
  if (!(getenv("WOQ_PERSIST_HINTS") && getenv("WOQ_PERSIST_HINTS")[0] == '1')) a.flags |= 16;
  if (getenv("WOQ_PERSIST_KNOCK")) a.flags |= (atoi(getenv("WOQ_PERSIST_KNOCK")) & 7) << 5;  // timing experiments
  if (getenv("WOQ_PERSIST_RING")) a.ring_tiles = std::max(2 * PS_D, std::min(a.ring_tiles, atoi(getenv("WOQ_PERSIST_RING")) & ~3));
  p->lds = (size_t)a.o_ring + (size_t)a.ring_tiles * 1024;
  p->grid = G;
  p->kv_dtype = d.kv_dtype;
  // device tables and vectors
  if (hipMalloc((void**)&p->d_layers, sizeof(PsLayer) * d.layers) != hipSuccess ||
      hipMemcpy(p->d_layers, hl.data(), sizeof(PsLayer) * d.layers, hipMemcpyHostToDevice) != hipSuccess) {
    persist_destroy(p);
    return no("device allocation");
  }
  const int attn_k = d.heads * d.head_dim;
  const size_t nb_h = d.hidden / 16, nb_a = attn_k / 16, nb_c = d.inter / 16;
  const size_t words = (nb_h + nb_a + nb_c) * 9 + (size_t)4 * G / 2 + 8;
  if (hipMalloc(&p->d_vecs, words * 8) != hipSuccess || hipMemset(p->d_vecs, 0, words * 8) != hipSuccess) {
    persist_destroy(p);
    return no("device allocation");
  }
  unsigned long long* v = (unsigned long long*)p->d_vecs;
  a.xg_hidden = XgVec{v, v + nb_h * 8}, v += nb_h * 9;
  a.xg_attn = XgVec{v, v + nb_a * 8}, v += nb_a * 9;
  a.xg_act = XgVec{v, v + nb_c * 8}, v += nb_c * 9;
  a.hint = (unsigned int*)v;
  a.layers = p->d_layers;
  a.seq = d.seq;
  a.pos = d.pos;
  a.status = d.status;
  a.cs = d.cs;
  a.sn = d.sn;
  a.x0 = d.x0;
  a.ssq0 = d.ssq0;
  a.qkv_g = d.qkv_g;
  a.hidden_buf = d.hidden_buf;
  a.stamps = nullptr;
  return p;
}

void persist_rebind(Persist* p, const int32_t* pos, float* hidden_buf) {
  p->args.pos = pos;
  p->args.hidden_buf = hidden_buf;
}
int persist_ring_tiles(const Persist* p) { return p ? p->args.ring_tiles : 0; }
int persist_grid(const Persist* p) { return p ? p->grid : 0; }
void persist_set_stamps(Persist* p, unsigned long long* dev) { p->args.stamps = dev; }

template <typename KV, int SMODE, bool ASYM>
static int launch_persist_t(Persist* p, hipStream_t st) {
  auto kern = persist_kernel<KV, SMODE, ASYM>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p->grid), dim3(PS_THREADS), p->lds, st, p->args);
  return 0;
}

// all layers of one decode step: in = the embedding launch's outputs, out = hidden_buf (the head launch's input)
template <typename KV>
static int launch_persist_kv(Persist* p, hipStream_t st) {
  const bool sm1 = (p->args.flags & 2) != 0, asym = (p->args.flags & 4) != 0;
  if (!sm1) return asym ? launch_persist_t<KV, 0, true>(p, st) : launch_persist_t<KV, 0, false>(p, st);
  return asym ? launch_persist_t<KV, 1, true>(p, st) : launch_persist_t<KV, 1, false>(p, st);
}
int persist_launch(Persist* p, hipStream_t st) {
  if (p->kv_dtype == WOQ_F16) return launch_persist_kv<_Float16>(p, st);
  if (p->kv_dtype == WOQ_FP8_E4M3) return launch_persist_kv<Fp8>(p, st);
  return launch_persist_kv<__bf16>(p, st);
}

}  // namespace woq
