// pdk_probe.hip — feasibility probe (development tool, not product): ONE persistent launch per token for a
// Llama-2-7B-shaped int4 decoder — 256 workgroups (one per CU), 8 streaming waves + 1 communicator wave each, phases
// chained by data-tagged 8-byte granules + sharded arrival counters instead of kernel boundaries, next phase's first
// weight tiles prefetched into registers before the hand-off wait. Synthetic weights, dummy attention: measures what
// the structure can reach before it is built for real.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/pdk_probe.hip -o tools/pdk_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../intel_extension_for_transformers_amd/csrc/woq_device.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

using namespace woq;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int NSW = 8;          // streaming waves
constexpr int NTH = NSW * 64;  // wave 0 doubles as the communicator
constexpr int AUX_NT = 2, AUX_SC1 = 16;
constexpr int MAXSTRIP = 6;     // strips a workgroup owns in one phase (3 gate/up pairs)

__device__ __forceinline__ rsrc_t mk(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float limb_combine(const i32x4& d) {
  const int i0 = d.x + (d.w << 7), i1 = d.y + (d.w << 7), i2 = d.z - (d.w << 6);
  return fmaf((float)i2, 65536.f, fmaf((float)i1, 256.f, (float)i0));
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct LayerW {
  const u32x4* q[4];      // qkv, o, gate_up, down: qdata
  const uint16_t* sc[4];  // fp16 scales [tiles_n][n_groups][16]
  const float* ln1;
  const float* ln2;
};
struct PdkArgs {
  const LayerW* layers;
  int n_layers, tkh, tki, n_qkv_strips, n_pairs;  // tiles_k(hidden), tiles_k(inter), strips of qkv, gate/up pairs
  u64* H;     // granules [hidden]
  u64* QKV;   // [qkv_n]
  u64* ATT;   // [hidden]
  u64* ACT;   // [inter]
  unsigned* cnt;  // [4 buffers][8 shards][16 words]
  unsigned* err;
  unsigned* serial;
  unsigned long long* dbg;  // [layers][16] wall-clock stamps of workgroup `dbg_wg`'s communicator
  int dbg_wg;
  float eps;
  int hidden, inter, heads;
};

template <int TPW, int CB>
struct WT {
  u32x4 w[CB][TPW];
  uint16_t sc[CB][TPW];
};

// all tile + scale loads of strip(s) [strip*CB, strip*CB+CB) for this wave's K slice [kt0, kt0+cnt)
template <int TPW, int CB>
__device__ __forceinline__ void load_strip(WT<TPW, CB>& W, const u32x4* q, const uint16_t* sc, int strip, int tiles_k,
                                           int kt0, int cnt, int lane, bool live) {
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int tn = strip * CB + cb;
    const rsrc_t rs = mk(sc + (size_t)tn * tiles_k * 16, live ? tiles_k * 32 : 0);
    const rsrc_t rq = mk(q + (size_t)tn * tiles_k * 64, live ? min(kt0 + cnt, tiles_k) * 1024 : 0);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
      W.sc[cb][t] = __builtin_amdgcn_raw_buffer_load_b16(rs, (lane & 15) * 2, min(kt0 + t, tiles_k - 1) * 32, 0);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
      W.w[cb][t] = __builtin_amdgcn_raw_buffer_load_b128(rq, lane * 16 + t * 1024, kt0 * 1024, AUX_NT);
  }
}

template <int TPW, int CB>
__device__ __forceinline__ void consume_strip(const WT<TPW, CB>& W, const unsigned char* a_base, int a_step_t,
                                              int a_step_h, float (&tot)[CB]) {
  const i32x4 izero = {0, 0, 0, 0};
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) tot[cb] = 0.f;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const i32x4 a0 = *(const i32x4*)(a_base + t * a_step_t);
    const i32x4 a1 = *(const i32x4*)(a_base + t * a_step_t + a_step_h);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const u32x4 wv = W.w[cb][t];
      const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u), (int)((wv.y << 4) & 0xf0f0f0f0u),
                        (int)(wv.y & 0xf0f0f0f0u)};
      const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u), (int)((wv.w << 4) & 0xf0f0f0f0u),
                        (int)(wv.w & 0xf0f0f0f0u)};
      i32x4 d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, izero, 0, 0, 0);
      d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, d, 0, 0, 0);
      tot[cb] = fmaf(f16_bits_to_f32(W.sc[cb][t]), limb_combine(d), tot[cb]);
    }
  }
}

// communicator: wait until `expected` producers have arrived on buffer `buf` (sum of 8 shards), bounded
__device__ __forceinline__ void wait_count(const PdkArgs& a, int buf, unsigned expected, int lane) {
  const unsigned* c = a.cnt + buf * 128;
  for (int it = 0; it < 200000; ++it) {
    unsigned v = 0;
    if (lane < 8) v = __hip_atomic_load(c + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float s = wave_sum_dpp((float)v);  // < 2^24 arrivals in this probe
    if ((unsigned)s >= expected) return;
    if ((it & 63) == 63 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void arrive(const PdkArgs& a, int buf, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's granule stores are out
  if (lane == 0)
    __hip_atomic_fetch_add(a.cnt + buf * 128 + (blockIdx.x & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void put(u64* g, int i, unsigned tag, float v) {
  __hip_atomic_store(g + i, ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// streaming wave: fetch this wave's slice of a granule vector (tag-validated, retried), optional RMSNorm weight,
// convert to the three int8 limb rows in its LDS strip. Returns (sum of squares, 2^(e-25)) via references.
template <int TPW>
__device__ __forceinline__ void stage_slice(const PdkArgs& a, const u64* vec, unsigned tag, int kbase, int xlen,
                                            const float* g, unsigned char* strip, int RB, int lane, float& ss_out,
                                            float& unsc_out) {
  constexpr int XJ = TPW / 2;  // blocks of 256 elements: each lane 4 consecutive elements (2 x 16-B granule pairs)
  const rsrc_t rv = mk(vec + kbase, xlen * 8);
  float4_t xv[XJ];
  for (int attempt = 0; attempt < 4096; ++attempt) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const u32x4 g0 = __builtin_amdgcn_raw_buffer_load_b128(rv, lane * 32 + j * 2048, 0, AUX_SC1);
      const u32x4 g1 = __builtin_amdgcn_raw_buffer_load_b128(rv, lane * 32 + j * 2048 + 16, 0, AUX_SC1);
      const bool live = j * 256 + lane * 4 < xlen;
      ok &= !live || (g0.y == tag && g0.w == tag && g1.y == tag && g1.w == tag);
      xv[j] = (float4_t){__uint_as_float(g0.x), __uint_as_float(g0.z), __uint_as_float(g1.x), __uint_as_float(g1.z)};
      if (!live) xv[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    if (__all(ok)) {
      if (attempt > 0 && lane == 0) atomicAdd(a.err + 4, (unsigned)attempt);
      break;
    }
    if (attempt == 4095 && lane == 0) __hip_atomic_store(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_sleep(4);
  }
  float ss = 0.f, amax = 0.f;
  const rsrc_t rg = mk(g ? g + kbase : (const float*)vec, g ? xlen * 4 : 0);
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    ss = fmaf(xv[j].x, xv[j].x, fmaf(xv[j].y, xv[j].y, fmaf(xv[j].z, xv[j].z, fmaf(xv[j].w, xv[j].w, ss))));
    if (g) xv[j] = xv[j] * __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, lane * 16 + j * 1024, 0, 0));
    amax = fmaxf(fmaxf(amax, fabsf(xv[j].x)), fmaxf(fabsf(xv[j].y), fmaxf(fabsf(xv[j].z), fabsf(xv[j].w))));
  }
  amax = wave_max_dpp(amax);
  ss_out = wave_sum_dpp(ss);
  int e = 0;
  if (amax > 0.f && amax < INFINITY) e = max(-100, min(100, __builtin_amdgcn_frexp_expf(amax)));
  const float sfix = ldexpf(1.f, 21 - e);
  unsc_out = ldexpf(1.f, e - 25);
  unsigned char* r0 = strip + lane * 4;
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const uint32_t A = __float_as_uint(fmaf(xv[j].x, sfix, 12582912.f)), B = __float_as_uint(fmaf(xv[j].y, sfix, 12582912.f));
    const uint32_t C = __float_as_uint(fmaf(xv[j].z, sfix, 12582912.f)), D = __float_as_uint(fmaf(xv[j].w, sfix, 12582912.f));
    const uint32_t ab0 = __builtin_amdgcn_perm(B, A, 0x0c0c0400u), cd0 = __builtin_amdgcn_perm(D, C, 0x04000c0cu);
    const uint32_t ab1 = __builtin_amdgcn_perm(B, A, 0x0c0c0501u), cd1 = __builtin_amdgcn_perm(D, C, 0x05010c0cu);
    const uint32_t ab2 = __builtin_amdgcn_perm(B, A, 0x0c0c0602u), cd2 = __builtin_amdgcn_perm(D, C, 0x06020c0cu);
    *(uint32_t*)(r0 + j * 256) = (ab0 | cd0) ^ 0x80808080u;
    *(uint32_t*)(r0 + RB + j * 256) = (ab1 | cd1) ^ 0x80808080u;
    *(uint32_t*)(r0 + 2 * RB + j * 256) = ab2 | cd2;
  }
  __builtin_amdgcn_wave_barrier();
}

#define WSTAMP(k)                                                                        \
  do {                                                                                   \
    if (wid == 3 && b == a.dbg_wg && lane == 0) a.dbg[(a.n_layers + l) * 16 + (k)] = wall_clock64(); \
  } while (0)
#define STAMP(k)                                                          \
  do {                                                                     \
    if (comm && b == a.dbg_wg && lane == 0) a.dbg[l * 16 + (k)] = wall_clock64(); \
  } while (0)

template <int TPWH, int TPWI>
__global__ __launch_bounds__(NTH) void pdk_kernel(PdkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int TPWM = TPWH > TPWI ? TPWH : TPWI;
  constexpr int RB = TPWM * 128 + 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, G = gridDim.x;
  unsigned char* zero_blk = sm + wid * 512;
  unsigned char* ones_blk = zero_blk + 256;
  unsigned char* strips = sm + NSW * 512;
  unsigned char* strip = strips + (size_t)wid * 3 * RB;
  float* slab = (float*)(strips + (size_t)NSW * 3 * RB);  // [NSW][MAXSTRIP][16]
  float* ssq = slab + NSW * MAXSTRIP * 16;                  // [NSW]
  ((uint32_t*)zero_blk)[lane] = 0u;
  ((uint32_t*)ones_blk)[lane] = 0x01010101u;
  const bool comm = wid == 0;
  const int i16 = lane & 15, kq = lane >> 4;
  // A-operand lane addresses (batch 1: rows 0..2 limbs, row 3 ones, the rest zero)
  const unsigned char* a_base = i16 < 3 ? strip + (size_t)i16 * RB + kq * 16 : (i16 == 3 ? ones_blk + kq * 16 : zero_blk + kq * 16);
  const int a_st = i16 < 3 ? 128 : 0, a_sh = i16 < 3 ? 64 : 0;
  // K slices: wave 0 (the communicator) takes a small share because it cannot prefetch across a hand-off (its polls
  // would queue behind its own loads); waves 1..7 split the rest evenly
  auto slice = [&](int tiles, int share0, int& kt0, int& cnt) {
    if (wid == 0) {
      kt0 = 0;
      cnt = share0;
    } else {
      const int rest = tiles - share0, base = rest / (NSW - 1), rem = rest % (NSW - 1), w = wid - 1;
      kt0 = share0 + w * base + min(w, rem);
      cnt = base + (w < rem ? 1 : 0);
    }
  };
  int kt0h, cnth, kt0i, cnti;
  slice(a.tkh, a.tkh / 8, kt0h, cnth);
  slice(a.tki, a.tki / 16, kt0i, cnti);
  const unsigned serial = a.serial[0];
  const unsigned per_token = 5u * a.n_layers + 1u;
  unsigned ep = serial * per_token + 1u;                 // tag of the embedding-phase H
  unsigned n_h = serial * (2u * a.n_layers + 1u) * G;    // arrivals seen so far on each buffer's counter
  unsigned n_qkv = serial * a.n_layers * G, n_att = serial * a.n_layers * a.heads, n_act = serial * a.n_layers * G;
  const int my_pairs = (a.n_pairs - b + G - 1) / G;      // gate/up pairs of this workgroup
  float h_own = 0.f;                                      // communicator lanes 0..15: this workgroup's H strip

  // ---- embedding stand-in: publish H strip ----
  if (comm) {
    if (lane < 16) {
      h_own = 0.01f * (float)((b * 16 + lane) % 97 - 48);
      put(a.H, b * 16 + lane, ep, h_own);
    }
    arrive(a, 0, lane);
  }
  n_h += G;
  const bool s1 = b + G < a.n_qkv_strips, s2 = b + 2 * G < a.n_qkv_strips;
  float ss = 0.f, unsc = 0.f;

  if (wid != 0) {
    // ================================ streaming waves ================================
    // Every weight structure is loaded right AFTER the partials barrier of the previous phase (so the communicator's
    // publish never waits behind this wave's issue stall) and consumed after the next hand-off.
    WT<TPWH, 1> Wq, Wq1, Wq2;
    {
      const LayerW L0 = a.layers[0];
      load_strip<TPWH, 1>(Wq, L0.q[0], L0.sc[0], b, a.tkh, kt0h, cnth, lane, true);
      load_strip<TPWH, 1>(Wq1, L0.q[0], L0.sc[0], b + G, a.tkh, kt0h, cnth, lane, s1);
      load_strip<TPWH, 1>(Wq2, L0.q[0], L0.sc[0], b + 2 * G, a.tkh, kt0h, cnth, lane, s2);
    }
    for (int l = 0; l < a.n_layers; ++l) {
      const LayerW L = a.layers[l];
      float tot[1], tot2[2];
      // ---- qkv ----
      wg_barrier();
      WSTAMP(0);
      stage_slice<TPWH>(a, a.H, ep, kt0h * 128, cnth * 128, L.ln1, strip, RB, lane, ss, unsc);
      WSTAMP(1);
      if (lane == 0) ssq[wid] = ss;
      consume_strip<TPWH, 1>(Wq, a_base, a_st, a_sh, tot);
      WSTAMP(2);
      if (lane < 16) slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot[0] * unsc;
      consume_strip<TPWH, 1>(Wq1, a_base, a_st, a_sh, tot);
      if (lane < 16) slab[(wid * MAXSTRIP + 1) * 16 + lane] = tot[0] * unsc;
      consume_strip<TPWH, 1>(Wq2, a_base, a_st, a_sh, tot);
      if (lane < 16) slab[(wid * MAXSTRIP + 2) * 16 + lane] = tot[0] * unsc;
      WSTAMP(3);
      wg_barrier();
      ++ep;
      WT<TPWH, 1> Wo;
      load_strip<TPWH, 1>(Wo, L.q[1], L.sc[1], b, a.tkh, kt0h, cnth, lane, true);
      ++ep;
      // ---- o ----
      wg_barrier();
      stage_slice<TPWH>(a, a.ATT, ep, kt0h * 128, cnth * 128, nullptr, strip, RB, lane, ss, unsc);
      consume_strip<TPWH, 1>(Wo, a_base, a_st, a_sh, tot);
      if (lane < 16) slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot[0] * unsc;
      wg_barrier();
      ++ep;
      WT<TPWH, 2> Wg0, Wg1;
      load_strip<TPWH, 2>(Wg0, L.q[2], L.sc[2], b, a.tkh, kt0h, cnth, lane, my_pairs > 0);
      load_strip<TPWH, 2>(Wg1, L.q[2], L.sc[2], b + G, a.tkh, kt0h, cnth, lane, my_pairs > 1);
      // ---- gate/up ----
      wg_barrier();
      stage_slice<TPWH>(a, a.H, ep, kt0h * 128, cnth * 128, L.ln2, strip, RB, lane, ss, unsc);
      if (lane == 0) ssq[wid] = ss;
      {
        WT<TPWH, 2> W2;
        load_strip<TPWH, 2>(W2, L.q[2], L.sc[2], b + 2 * G, a.tkh, kt0h, cnth, lane, my_pairs > 2);
        consume_strip<TPWH, 2>(Wg0, a_base, a_st, a_sh, tot2);
        if (lane < 16) {
          slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot2[0] * unsc;
          slab[(wid * MAXSTRIP + 1) * 16 + lane] = tot2[1] * unsc;
        }
        consume_strip<TPWH, 2>(Wg1, a_base, a_st, a_sh, tot2);
        if (lane < 16) {
          slab[(wid * MAXSTRIP + 2) * 16 + lane] = tot2[0] * unsc;
          slab[(wid * MAXSTRIP + 3) * 16 + lane] = tot2[1] * unsc;
        }
        consume_strip<TPWH, 2>(W2, a_base, a_st, a_sh, tot2);
        if (lane < 16) {
          slab[(wid * MAXSTRIP + 4) * 16 + lane] = tot2[0] * unsc;
          slab[(wid * MAXSTRIP + 5) * 16 + lane] = tot2[1] * unsc;
        }
      }
      wg_barrier();
      ++ep;
      WT<TPWI, 1> Wd;
      load_strip<TPWI, 1>(Wd, L.q[3], L.sc[3], b, a.tki, kt0i, cnti, lane, true);
      // ---- down ----
      wg_barrier();
      stage_slice<TPWI>(a, a.ACT, ep, kt0i * 128, min(cnti * 128, a.inter - kt0i * 128), nullptr, strip, RB, lane, ss, unsc);
      consume_strip<TPWI, 1>(Wd, a_base, a_st, a_sh, tot);
      if (lane < 16) slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot[0] * unsc;
      wg_barrier();
      ++ep;
      {
        const bool more = l + 1 < a.n_layers;
        const LayerW Ln = a.layers[more ? l + 1 : l];
        load_strip<TPWH, 1>(Wq, Ln.q[0], Ln.sc[0], b, a.tkh, kt0h, cnth, lane, more);
        load_strip<TPWH, 1>(Wq1, Ln.q[0], Ln.sc[0], b + G, a.tkh, kt0h, cnth, lane, more && s1);
        load_strip<TPWH, 1>(Wq2, Ln.q[0], Ln.sc[0], b + 2 * G, a.tkh, kt0h, cnth, lane, more && s2);
      }
    }
  } else {
    // ================================ communicator wave (also streams a K share, without prefetch) ================
    for (int l = 0; l < a.n_layers; ++l) {
      const LayerW L = a.layers[l];
      float tot[1], tot2[2];
      // ---- qkv ----
      STAMP(0);
      wait_count(a, 0, n_h, lane);
      STAMP(1);
      wg_barrier();
      {
        WT<TPWH, 1> Wq, Wq1, Wq2;
        load_strip<TPWH, 1>(Wq, L.q[0], L.sc[0], b, a.tkh, kt0h, cnth, lane, true);
        load_strip<TPWH, 1>(Wq1, L.q[0], L.sc[0], b + G, a.tkh, kt0h, cnth, lane, s1);
        load_strip<TPWH, 1>(Wq2, L.q[0], L.sc[0], b + 2 * G, a.tkh, kt0h, cnth, lane, s2);
        stage_slice<TPWH>(a, a.H, ep, kt0h * 128, cnth * 128, L.ln1, strip, RB, lane, ss, unsc);
        if (lane == 0) ssq[wid] = ss;
        consume_strip<TPWH, 1>(Wq, a_base, a_st, a_sh, tot);
        if (lane < 16) slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot[0] * unsc;
        consume_strip<TPWH, 1>(Wq1, a_base, a_st, a_sh, tot);
        if (lane < 16) slab[(wid * MAXSTRIP + 1) * 16 + lane] = tot[0] * unsc;
        consume_strip<TPWH, 1>(Wq2, a_base, a_st, a_sh, tot);
        if (lane < 16) slab[(wid * MAXSTRIP + 2) * 16 + lane] = tot[0] * unsc;
      }
      wg_barrier();
      STAMP(2);
      ++ep;
      {
        float sq = 0.f;
        for (int w2 = 0; w2 < NSW; ++w2) sq += ssq[w2];
        const float inv = 1.0f / sqrtf(sq / (float)a.hidden + a.eps);
        if (lane < 48) {
          const int s = lane >> 4;
          float v = 0.f;
          for (int w2 = 0; w2 < NSW; ++w2) v += slab[(w2 * MAXSTRIP + s) * 16 + (lane & 15)];
          if (b + s * G < a.n_qkv_strips) put(a.QKV, (b + s * G) * 16 + (lane & 15), ep, v * inv);
        }
        arrive(a, 1, lane);
      }
      n_qkv += G;
      STAMP(3);
      // ---- attention stand-in (workgroups < heads) ----
      ++ep;
      if (b < a.heads) {
        wait_count(a, 1, n_qkv, lane);
        const int hd = a.hidden / a.heads;
        for (int i = lane; i < hd; i += 64) {
          u64 g = 0;
          for (int it = 0; it < 100000; ++it) {
            g = __hip_atomic_load(a.QKV + b * hd + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(g >> 32) == ep - 1) break;
          }
          put(a.ATT, b * hd + i, ep, 0.5f * __uint_as_float((unsigned)g));
        }
        arrive(a, 2, lane);
      }
      n_att += a.heads;
      // ---- o ----
      STAMP(4);
      wait_count(a, 2, n_att, lane);
      STAMP(5);
      wg_barrier();
      {
        WT<TPWH, 1> Wo;
        load_strip<TPWH, 1>(Wo, L.q[1], L.sc[1], b, a.tkh, kt0h, cnth, lane, true);
        stage_slice<TPWH>(a, a.ATT, ep, kt0h * 128, cnth * 128, nullptr, strip, RB, lane, ss, unsc);
        consume_strip<TPWH, 1>(Wo, a_base, a_st, a_sh, tot);
        if (lane < 16) slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot[0] * unsc;
      }
      wg_barrier();
      ++ep;
      if (lane < 16) {
        float v = 0.f;
        for (int w2 = 0; w2 < NSW; ++w2) v += slab[(w2 * MAXSTRIP + 0) * 16 + lane];
        h_own += v;
        put(a.H, b * 16 + lane, ep, h_own);
      }
      arrive(a, 0, lane);
      n_h += G;
      STAMP(6);
      // ---- gate/up ----
      wait_count(a, 0, n_h, lane);
      STAMP(7);
      wg_barrier();
      {
        WT<TPWH, 2> Wg0, Wg1, W2;
        load_strip<TPWH, 2>(Wg0, L.q[2], L.sc[2], b, a.tkh, kt0h, cnth, lane, my_pairs > 0);
        load_strip<TPWH, 2>(Wg1, L.q[2], L.sc[2], b + G, a.tkh, kt0h, cnth, lane, my_pairs > 1);
        load_strip<TPWH, 2>(W2, L.q[2], L.sc[2], b + 2 * G, a.tkh, kt0h, cnth, lane, my_pairs > 2);
        stage_slice<TPWH>(a, a.H, ep, kt0h * 128, cnth * 128, L.ln2, strip, RB, lane, ss, unsc);
        if (lane == 0) ssq[wid] = ss;
        consume_strip<TPWH, 2>(Wg0, a_base, a_st, a_sh, tot2);
        if (lane < 16) {
          slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot2[0] * unsc;
          slab[(wid * MAXSTRIP + 1) * 16 + lane] = tot2[1] * unsc;
        }
        consume_strip<TPWH, 2>(Wg1, a_base, a_st, a_sh, tot2);
        if (lane < 16) {
          slab[(wid * MAXSTRIP + 2) * 16 + lane] = tot2[0] * unsc;
          slab[(wid * MAXSTRIP + 3) * 16 + lane] = tot2[1] * unsc;
        }
        consume_strip<TPWH, 2>(W2, a_base, a_st, a_sh, tot2);
        if (lane < 16) {
          slab[(wid * MAXSTRIP + 4) * 16 + lane] = tot2[0] * unsc;
          slab[(wid * MAXSTRIP + 5) * 16 + lane] = tot2[1] * unsc;
        }
      }
      wg_barrier();
      ++ep;
      {
        float sq = 0.f;
        for (int w2 = 0; w2 < NSW; ++w2) sq += ssq[w2];
        const float inv = 1.0f / sqrtf(sq / (float)a.hidden + a.eps);
        if (lane < 48) {
          const int p = lane >> 4;
          float gte = 0.f, up = 0.f;
          for (int w2 = 0; w2 < NSW; ++w2) {
            gte += slab[(w2 * MAXSTRIP + 2 * p) * 16 + (lane & 15)];
            up += slab[(w2 * MAXSTRIP + 2 * p + 1) * 16 + (lane & 15)];
          }
          gte *= inv;
          up *= inv;
          if (p < my_pairs) put(a.ACT, (b + p * G) * 16 + (lane & 15), ep, gte / (1.0f + __expf(-gte)) * up);
        }
        arrive(a, 3, lane);
      }
      n_act += G;
      STAMP(8);
      // ---- down ----
      wait_count(a, 3, n_act, lane);
      STAMP(9);
      wg_barrier();
      {
        WT<TPWI, 1> Wd;
        load_strip<TPWI, 1>(Wd, L.q[3], L.sc[3], b, a.tki, kt0i, cnti, lane, true);
        stage_slice<TPWI>(a, a.ACT, ep, kt0i * 128, min(cnti * 128, a.inter - kt0i * 128), nullptr, strip, RB, lane, ss, unsc);
        consume_strip<TPWI, 1>(Wd, a_base, a_st, a_sh, tot);
        if (lane < 16) slab[(wid * MAXSTRIP + 0) * 16 + lane] = tot[0] * unsc;
      }
      wg_barrier();
      ++ep;
      if (lane < 16) {
        float v = 0.f;
        for (int w2 = 0; w2 < NSW; ++w2) v += slab[(w2 * MAXSTRIP + 0) * 16 + lane];
        h_own += v;
        put(a.H, b * 16 + lane, ep, h_own);
      }
      arrive(a, 0, lane);
      n_h += G;
      STAMP(10);
    }
  }
  if (comm && b == 0 && lane == 0) a.serial[0] = serial + 1;
}

__global__ void fill_random(unsigned* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    p[i] = x;
  }
}
__global__ void fill_u16(unsigned short* p, size_t n, unsigned short v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_f32(float* p, size_t n, float v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 32;
  const int tokens = argc > 2 ? atoi(argv[2]) : 20;
  const int hidden = 4096, inter = 11008, heads = 32;
  const int tkh = hidden / 128, tki = (inter + 127) / 128;
  const int qkv_n = 3 * hidden;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  printf("CUs %d\n", G);
  std::vector<LayerW> hl(layers);
  const int Ns[4] = {qkv_n, hidden, 2 * inter, hidden};
  const int tks[4] = {tkh, tkh, tkh, tki};
  double bytes_per_token = 0;
  for (int l = 0; l < layers; ++l) {
    for (int j = 0; j < 4; ++j) {
      const size_t qb = (size_t)(Ns[j] / 16) * tks[j] * 1024, sb = (size_t)(Ns[j] / 16) * tks[j] * 32;
      unsigned char* p;
      CK(hipMalloc(&p, qb + sb));
      fill_random<<<1024, 256, 0, st>>>((unsigned*)p, qb / 4, 77u + l * 4 + j);
      fill_u16<<<256, 256, 0, st>>>((unsigned short*)(p + qb), sb / 2, 0x1400);  // ~1e-3
      hl[l].q[j] = (const u32x4*)p;
      hl[l].sc[j] = (const uint16_t*)(p + qb);
      bytes_per_token += (double)qb + sb;
    }
    float* ln;
    CK(hipMalloc(&ln, 2 * hidden * 4));
    fill_f32<<<64, 256, 0, st>>>(ln, 2 * hidden, 1.0f);
    hl[l].ln1 = ln;
    hl[l].ln2 = ln + hidden;
  }
  PdkArgs a;
  LayerW* dl;
  CK(hipMalloc(&dl, layers * sizeof(LayerW)));
  CK(hipMemcpy(dl, hl.data(), layers * sizeof(LayerW), hipMemcpyHostToDevice));
  a.layers = dl;
  a.n_layers = layers;
  a.tkh = tkh;
  a.tki = tki;
  a.n_qkv_strips = qkv_n / 16;
  a.n_pairs = inter / 16;
  CK(hipMalloc(&a.H, hidden * 8));
  CK(hipMalloc(&a.QKV, qkv_n * 8));
  CK(hipMalloc(&a.ATT, hidden * 8));
  CK(hipMalloc(&a.ACT, inter * 8));
  CK(hipMemset(a.H, 0, hidden * 8));
  CK(hipMemset(a.QKV, 0, qkv_n * 8));
  CK(hipMemset(a.ATT, 0, hidden * 8));
  CK(hipMemset(a.ACT, 0, inter * 8));
  CK(hipMalloc(&a.cnt, 4 * 128 * 4));
  CK(hipMemset(a.cnt, 0, 4 * 128 * 4));
  CK(hipMalloc(&a.err, 64));
  CK(hipMemset(a.err, 0, 64));
  CK(hipMalloc(&a.serial, 64));
  CK(hipMemset(a.serial, 0, 64));
  CK(hipMalloc(&a.dbg, 2 * layers * 16 * 8));
  CK(hipMemset(a.dbg, 0, 2 * layers * 16 * 8));
  a.dbg_wg = argc > 3 ? atoi(argv[3]) : 100;
  a.eps = 1e-5f;
  a.hidden = hidden;
  a.inter = inter;
  a.heads = heads;
  constexpr int TPWH = 4, TPWI = 12;
  const int RB = 12 * 128 + 16;
  const size_t lds = NSW * 512 + (size_t)NSW * 3 * RB + NSW * MAXSTRIP * 16 * 4 + NSW * 4 + 64;
  auto kern = pdk_kernel<TPWH, TPWI>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(NTH), lds, st, a);
  CK(hipStreamSynchronize(st));
  unsigned err = 0;
  CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
  printf("warmup err flag %u\n", err);
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < tokens; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(NTH), lds, st, a);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> hh(16);
  CK(hipMemcpy(hh.data(), a.H, 16 * 8, hipMemcpyDeviceToHost));
  const double us = ms * 1e3 / tokens;
  printf("layers %d: %.1f us/token (%.2f us/layer), %.0f GB/s on %.1f MB/token, err %u, H[0]=%g tag %llu\n", layers, us,
         us / layers, bytes_per_token / us / 1e3, bytes_per_token / 1e6, err,
         (double)__builtin_bit_cast(float, (unsigned)hh[0]), hh[0] >> 32);
  std::vector<unsigned long long> dbg(2 * layers * 16);
  CK(hipMemcpy(dbg.data(), a.dbg, 2 * layers * 16 * 8, hipMemcpyDeviceToHost));
  {
    unsigned retries = 0;
    CK(hipMemcpy(&retries, a.err + 4, 4, hipMemcpyDeviceToHost));
    double w[3] = {0, 0, 0};
    for (int l = 1; l < layers; ++l)
      for (int k = 0; k < 3; ++k) w[k] += (double)(dbg[(layers + l) * 16 + k + 1] - dbg[(layers + l) * 16 + k]) * 0.01;
    printf("wave 3 qkv phase: granules+stage %.2f us, first strip %.2f us, strips 2-3 %.2f us; granule retries total %u\n",
           w[0] / (layers - 1), w[1] / (layers - 1), w[2] / (layers - 1), retries);
  }
  const char* nm[10] = {"wait H (qkv)", "stage+compute qkv", "publish qkv", "(attn others)", "wait ATT", "stage+compute+publish o",
                        "wait H (gu)", "stage+compute+publish gu", "wait ACT", "stage+compute+publish down"};
  double acc[10] = {0};
  for (int l = 1; l < layers; ++l)
    for (int k = 0; k < 10; ++k) acc[k] += (double)(dbg[l * 16 + k + 1] - dbg[l * 16 + k]) * 0.01;
  printf("workgroup %d communicator, mean us per layer:\n", a.dbg_wg);
  for (int k = 0; k < 10; ++k) printf("   %-28s %7.2f\n", nm[k], acc[k] / (layers - 1));
  return 0;
}
