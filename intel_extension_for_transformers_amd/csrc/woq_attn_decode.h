// woq_attn_decode.h — single-query (decode) attention of one head as a device function, shared by the standalone
// launch (woq_ops.hip: attn_decode_kernel) and the fused qkv-GEMV + attention launch (woq_gemv_attn.hip).
// Reference: stock HF eager attention over the KV cache run by PyTorch CPU ops.
#pragma once
#include "woq_attn_merge.h"
#include "woq_device.h"
#include "woq_xq.h"

// polling interval of a wave that re-reads its granules, in units of 64 clocks (A/B builds: r06ap)
#ifndef WOQ_ATTN_POLL_SLEEP
#define WOQ_ATTN_POLL_SLEEP 2
#endif

namespace woq {

// Single-query attention for one new token: one workgroup (4 waves) per query head.
//   qkv: fp32 [(heads + 2*kv_heads) * HD] un-rotated projections of the new token.
//   RoPE is applied here to q and to the new k; rotated k and v are appended to the cache (by the first query
//   head of each kv group) at position pos. kv caches: [max_ctx, kv_heads, HD] (fp16 | bf16 | e4m3). out fp32 [heads*HD].
// Round 3 form — the launch is a latency chain (32 workgroups, a few dozen cached positions in the benchmark: 5.2 us
// per layer, of which a lone thread's 128-step dot product for the new position and twelve LDS-crossbar shuffles were
// more than a third), so the chain is what is cut:
//   * every wave is its own flash-decoding slice: it rotates q itself (wave-private LDS copy, no workgroup barrier),
//     owns the cached positions 64 i + 16 w + r (r < 16), and runs scores -> wave max -> exp / wave sum -> P.V over
//     them with DPP reductions only; the 4 (max, sum, o[HD]) partials meet behind the launch's ONE barrier;
//   * the new position's score is a wave-0 register dot product (the rotated q and k are already there), its
//     probability rides in wave 0's list;
//   * the first K rows and V rows of every wave depend only on `pos`: they are requested before anything else.
// SPLIT (long contexts, flash-decoding across workgroups): gridDim.y workgroups per head, each over its own slice of
// the cached positions (the last slice also takes the new position) -> un-normalised partial (o[HD], max, sum) per
// (head, slice) in `out`; attn_combine_kernel merges them. A lone workgroup per head streams the cache at one CU's
// ~10 B/clk, which is why contexts beyond a few thousand positions need the slices.
// LDS (floats): [4][HD] q copies | [HD] new v | [4][HD] partial outputs | [12] (4 max, 4 sum, new position's score) |
// [4][spw] scores
__host__ __device__ inline int attn_dec_spw(int span) { return ((span + 63) / 64) * 16 + 16; }
__host__ __device__ inline size_t attn_dec_lds_floats(int HD, int span) {
  return (size_t)9 * HD + 12 + 4 * (size_t)attn_dec_spw(span);
}

// Where q / k / v of the new token come from.
//   AttnPlain  : fp32 array written by an earlier launch; every wave reads and rotates q itself (no barrier).
//   AttnGranule: 8-byte {tag, fp32} granules that OTHER workgroups of the SAME launch are still writing with
//                write-through agent-scope stores (woq_gemv_attn.hip). Wave 0 re-reads its six granules per lane until
//                every tag is the expected one (cdna_hip_programming.md Guideline 16, form R2: the data is the flag —
//                no fence, no separate flag that could overtake its payload), rotates, and shares q through LDS behind
//                one workgroup barrier. The wait is bounded: after ~20 ms it raises *status and goes on.
struct AttnPlain {
  static constexpr bool granules = false;
  const float* qkv;
};
struct AttnGranule {
  static constexpr bool granules = true;
  const unsigned long long* g;
  unsigned int tag;
  int* status;
  unsigned long long* part_g = nullptr;  // SPLIT: the slices' partials as {tag, fp32} granules (attn_part_granule)
};
// SPLIT on a granule source (round 6): no combine launch — the slices of a head publish tagged partial granules
// (part_g) and finalise the head's XQ blocks themselves (woq_attn_merge.h, all-to-all merge; at most 16 slices here).
constexpr int ATTN_A2A_MAX_SLICES = 16;

// Who runs the body. AttnWgEnv: a 256-thread workgroup of its own (the standalone and the fused launches) — thread ids
// are the workgroup's, the two meeting points are workgroup barriers, the result leaves as fp32 (+ its XQ block).
// The persistent token kernel (woq_persist.hip) runs the body on four of its consumer waves with its own environment:
// sub-workgroup meeting points on an LDS counter (its loader wave never reaches an s_barrier) and tagged granules out.
struct AttnWgEnv {
  float* out;
  XqPtrs xo;
  XqPub pub;
  __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  // value `idx` of the attention output; called by whole 16-lane rows (idx & 15 == lane & 15)
  __device__ __forceinline__ void put(float v, int idx) const {
    out[idx] = v;
    if (xo.limbs != nullptr) {  // the o_proj GEMV's XQ input: one block per 16-lane row
      if (pub.flag != nullptr)  // an o_proj workgroup of the SAME launch may be waiting for this block
        xq_emit16<true>(v, xo, idx >> 4, idx & 15, pub);
      else
        xq_emit16<false>(v, xo, idx >> 4, idx & 15, pub);
    }
  }
};

// `h`: query head of this workgroup, `slice` / `n_slices`: its part of the cached positions (SPLIT), `sm`: LDS base
template <typename KV, int HD, bool SPLIT, typename SRC, typename ENV>
__device__ __forceinline__ void attn_decode_env(float* sm, int h, int slice, int n_slices, const SRC& src,
                                                KV* __restrict__ kcache, KV* __restrict__ vcache,
                                                const int32_t* __restrict__ pos_p, const float* __restrict__ cs,
                                                const float* __restrict__ sn, int heads, int kv_heads, int window,
                                                int spw, float* __restrict__ out, const ENV& env) {
  typedef typename KvVec8<KV>::type kv8;
  constexpr int half = HD / 2;
  constexpr int DPL = HD / 4;   // dims per lane in the score phase (4 lanes per position)
  constexpr int LPR = HD / 8;   // lanes per row in the P.V phase
  constexpr int GP = 64 / LPR;  // positions per wave per P.V pass
  const int tid = env.tid(), lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rep = heads / kv_heads, kh = h / rep;
  const int apos = pos_p[0];  // absolute position of the new token = number of cached positions
  // sliding window (HF Mistral `sliding_window`, 0 = none): the query sees positions [apos + 1 - window, apos]
  const int w_lo = window > 0 ? max(0, apos + 1 - window) : 0;
  int t_lo = w_lo, pos = apos - w_lo;  // this workgroup's cached slice is [t_lo, t_lo + pos)
  bool incl_new = true;
  if constexpr (SPLIT) {
    const int ns = n_slices, sp = slice;
    const int span = apos - w_lo;
    const int chunk = (((span + ns - 1) / ns) + 63) & ~63;
    t_lo = w_lo + min(sp * chunk, span);
    pos = min(apos - t_lo, chunk);
    incl_new = sp == ns - 1;
  }
  kcache += (size_t)t_lo * kv_heads * HD;
  vcache += (size_t)t_lo * kv_heads * HD;
  const int npos_abs = apos - t_lo;  // row of the new position relative to the re-based cache pointers
  float* qw = sm + (SRC::granules ? 0 : wid * HD);  // rotated q (pre-scaled by 1/sqrt(HD)): wave-private | shared
  float* vn = sm + 4 * HD;                 // new v, rounded to the cache dtype (wave 0 writes and reads it)
  float* slab = sm + 5 * HD;               // [4][HD] partial outputs
  float* ml = sm + 9 * HD;                 // [4] maxima, [4] sums
  float* scw = sm + 9 * HD + 12 + wid * spw;  // this wave's scores / probabilities, list order
  // this wave's cached positions: list index j <-> relative position 64 (j / 16) + 16 wid + (j % 16)
  const int n_w = 16 * (pos >> 6) + max(0, min((pos & 63) - 16 * wid, 16));
  const bool has_new = incl_new && wid == 0;
  // ---- requests that depend on `pos` only: K rows of the first score pass, V rows of the first P.V pass ----
  const int sub = lane & 3, r16 = lane >> 2;
  const int g = lane / LPR, l8 = lane % LPR;
  const int plast = max(pos - 1, 0);
  // Two request schedules. 16-bit caches: round 4's — two K passes and two V runs requested up front, the third V run
  // when the K registers die, double buffering in the loops (117 registers: four waves per SIMD, which the fused launch's
  // strips need). e4m3 cache (round 6): RINGS of register sets — pass / run i lives in set i mod depth and the set is
  // re-requested for i + depth as soon as it has been consumed; the sets are half the size, so four K passes and four V
  // runs are in flight (at long contexts a slice is a latency chain of its passes: Mistral-7B shape at 8k, eight per wave).
  constexpr bool kv8bit = sizeof(KV) == 1;
  constexpr int KD = 4, VD = 4;
  constexpr int VU = 16 / GP;  // P.V passes that cover one 16-position run
  auto kload = [&](int i, kv8 (&kv)[DPL / 8]) {
    const int tc = min(64 * i + 16 * wid + r16, plast);
    const kv8* kp = (const kv8*)(kcache + ((size_t)tc * kv_heads + kh) * HD + sub * DPL);
#pragma unroll
    for (int j = 0; j < DPL / 8; ++j) kv[j] = kp[j];
  };
  auto vload_run = [&](int r, kv8 (&vv)[VU]) {  // run r = list entries 16 r .. 16 r + 15 = positions 64 r + 16 wid + ...
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int tc = min(64 * r + 16 * wid + g + u * GP, plast);
      vv[u] = *(const kv8*)(vcache + ((size_t)tc * kv_heads + kh) * HD + l8 * 8);
    }
  };
  kv8 kr[kv8bit ? KD : 1][DPL / 8], vr[kv8bit ? VD : 1][VU];  // (the rings; one dead set for 16-bit caches)
  kv8 kpre[DPL / 8], kb[DPL / 8], vpre[VU], vpre2[VU];          // (round 4's sets; dead for an e4m3 cache)
  if constexpr (kv8bit) {
#pragma unroll
    for (int b2 = 0; b2 < KD; ++b2)
      if (b2 == 0 || n_w > 16 * b2) kload(b2, kr[b2]);
#pragma unroll
    for (int b2 = 0; b2 < VD; ++b2)
      if (b2 == 0 || n_w > 16 * b2) vload_run(b2, vr[b2]);
  } else {
    kload(0, kpre);
    if (n_w > 16) kload(1, kb);  // second score pass (positions 64 + 16 w + r): needs only `pos` as well
    vload_run(0, vpre);
    // round 4: the V rows of the SECOND 16-position run too (contexts 64 .. 128 used to pay one exposed round trip for
    // them behind the q hand-off, on the launch's tail: the 128-step bench value sat 4 % under the 20-step one)
    if (n_w > 16) vload_run(1, vpre2);
  }
  // ---- RoPE of q (every wave for itself | wave 0 for all). The new position — k rotated, k / v rounded to the cache
  // dtype and appended, its score — is wave 0's and enters the result as a FIFTH partial (max = its score, sum = 1,
  // output = its v) in the merge at the end, not as an entry of wave 0's list (round 6): with a granule source its q
  // strips land first and its k / v strips last (woq_gemv_attn.hip), so everything over the cache runs on q alone and
  // only `new_position` below stands behind the launch's last strips. ----
  const float scale = 1.0f / sqrtf((float)HD);
  float s_new = 0.f;
  float ra = 0.f, rb = 0.f, rc = 0.f, rs = 0.f;  // rotated, pre-scaled q halves and the cos / sin of this lane (lanes < half)
  const bool q_lane = lane < half && (!SRC::granules || wid == 0);
  if (q_lane) {
    rc = cs[(size_t)apos * half + lane], rs = sn[(size_t)apos * half + lane];
    const size_t oq = (size_t)h * HD + lane;
    float qa, qb;
    if constexpr (SRC::granules) {
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        const unsigned long long g0 = __hip_atomic_load(src.g + oq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long g1 = __hip_atomic_load(src.g + oq + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool good = (unsigned int)(g0 >> 32) == src.tag && (unsigned int)(g1 >> 32) == src.tag;
        qa = __uint_as_float((unsigned int)g0), qb = __uint_as_float((unsigned int)g1);
        if (__all(good)) break;  // every participating lane of the wave saw its tags
        if (wall_clock64() - t0 > 2000000ull) {  // 20 ms at 100 MHz: a producer is missing — say so, do not hang
          if (lane == 0) atomicOr(src.status, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(WOQ_ATTN_POLL_SLEEP);
      }
    } else {
      qa = src.qkv[oq], qb = src.qkv[oq + half];
    }
    // (explicit fused multiply-adds: hipcc contracts a * b + c * d by context, and the granule and plain sources of this
    // body are different contexts — the fused launch and the separate launches must stay bit-identical)
    ra = fmaf(qa, rc, -(qb * rs)) * scale, rb = fmaf(qb, rc, qa * rs) * scale;
    qw[lane] = ra;
    qw[lane + half] = rb;
  }
  // k / v of the new token (wave 0 of the workgroup that holds the new position): rotate k, round both through the cache
  // dtype, append (the first query head of each kv group), score against the rotated q still in this lane's registers
  auto new_position = [&]() {
    if (lane < half) {
      const size_t ok = (size_t)(heads + kh) * HD + lane, ov = (size_t)(heads + kv_heads + kh) * HD + lane;
      float ka, kb, va, vb;
      if constexpr (SRC::granules) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          const unsigned long long g2 = __hip_atomic_load(src.g + ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long g3 = __hip_atomic_load(src.g + ok + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long g4 = __hip_atomic_load(src.g + ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long g5 = __hip_atomic_load(src.g + ov + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool good = (unsigned int)(g2 >> 32) == src.tag && (unsigned int)(g3 >> 32) == src.tag &&
                            (unsigned int)(g4 >> 32) == src.tag && (unsigned int)(g5 >> 32) == src.tag;
          ka = __uint_as_float((unsigned int)g2), kb = __uint_as_float((unsigned int)g3);
          va = __uint_as_float((unsigned int)g4), vb = __uint_as_float((unsigned int)g5);
          if (__all(good)) break;
          if (wall_clock64() - t0 > 2000000ull) {
            if (lane == 0) atomicOr(src.status, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(WOQ_ATTN_POLL_SLEEP);
        }
      } else {
        ka = src.qkv[ok], kb = src.qkv[ok + half];
        va = src.qkv[ov], vb = src.qkv[ov + half];
      }
      const KV ka_r = (KV)fmaf(ka, rc, -(kb * rs)), kb_r = (KV)fmaf(kb, rc, ka * rs);
      const KV va_r = (KV)va, vb_r = (KV)vb;
      s_new = fmaf(ra, (float)ka_r, rb * (float)kb_r);
      vn[lane] = (float)va_r;
      vn[lane + half] = (float)vb_r;
      if (h % rep == 0) {
        KV* kd = kcache + ((size_t)npos_abs * kv_heads + kh) * HD;
        KV* vd = vcache + ((size_t)npos_abs * kv_heads + kh) * HD;
        kd[lane] = ka_r;
        kd[lane + half] = kb_r;
        vd[lane] = va_r;
        vd[lane + half] = vb_r;
      }
    }
    s_new = wave_sum_dpp(s_new);
    if (lane == 0) ml[8] = s_new;
  };
  if (wid == 0) {
    if (!has_new) {
      if (lane == 0) ml[8] = -INFINITY;  // this workgroup's slice does not hold the new position
    } else if constexpr (!SRC::granules) {
      new_position();  // its inputs are an earlier launch's: nothing to wait for
    }
  }
  if constexpr (SRC::granules)
    env.sync();  // q is wave 0's to give
  else
    __builtin_amdgcn_wave_barrier();
  // ---- scores of this wave's cached positions ----
  float qreg[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) qreg[i] = qw[sub * DPL + i];
  float lmax = -INFINITY;
  auto score = [&](int i, const kv8 (&kv)[DPL / 8]) {
    const int t = 64 * i + 16 * wid + r16;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < DPL / 8; ++j) {
      float kf[8];
      kv8_to_f32(kv[j], kf);
#pragma unroll
      for (int e = 0; e < 8; ++e) d = fmaf(qreg[j * 8 + e], kf[e], d);
    }
    d += WOQ_DPP_F32(d, 0xB1);  // quad_perm xor 1
    d += WOQ_DPP_F32(d, 0x4E);  // quad_perm xor 2: all four lanes of the position hold its score
    if (t < pos) {
      if (sub == 0) scw[16 * i + r16] = d;
      lmax = fmaxf(lmax, d);
    }
  };
  const int n_it = (n_w + 15) >> 4;  // 16-position passes of this wave
  if constexpr (kv8bit) {
    for (int i0 = 0; i0 < n_it; i0 += KD) {
#pragma unroll
      for (int b2 = 0; b2 < KD; ++b2) {
        const int i = i0 + b2;
        if (i < n_it) {
          score(i, kr[b2]);
          if (i + KD < n_it) kload(i + KD, kr[b2]);  // the set is free again: pass i + KD joins the ones in flight
        }
      }
    }
  } else if (n_it > 0) {
    score(0, kpre);
    // rows of pass i + 1 are in flight while pass i is scored
    for (int i = 1; i < n_it; i += 2) {
      if (i + 1 < n_it) kload(i + 1, kpre);
      score(i, kb);
      if (i + 1 < n_it) {
        if (i + 2 < n_it) kload(i + 2, kb);
        score(i + 1, kpre);
      }
    }
  }
  const float m_w = wave_max_dpp(lmax);
  const int n_l = n_w;  // entries of this wave's list (the new position is the merge's fifth partial)
  // round 4 (16-bit caches): the V rows of the third 16-position run are requested HERE — the K registers have just died,
  // and the exponentials below run under the request instead of in front of it (later runs are double-buffered in the loop)
  kv8 vnext[VU];
  if constexpr (!kv8bit) {
    if (n_w > 32) vload_run(2, vnext);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- probabilities and their sum ----
  float lsum = 0.f;
  for (int j = lane; j < n_l; j += 64) {
    const float p = __expf(scw[j] - m_w);
    scw[j] = p;
    lsum += p;
  }
  const float l_w = wave_sum_dpp(lsum);
  __builtin_amdgcn_wave_barrier();
  // ---- P.V over the wave's list: LPR lanes per row, GP rows per pass ----
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  auto pv = [&](int j, const kv8& vv) {
    const float p = j < n_w ? scw[j] : 0.f;
    float vf[8];
    kv8_to_f32(vv, vf);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(p, vf[i], acc[i]);
  };
  if constexpr (kv8bit) {
    for (int r0 = 0; 16 * r0 < n_w; r0 += VD) {  // runs in ascending order; run r + VD is requested when run r has been consumed
#pragma unroll
      for (int b2 = 0; b2 < VD; ++b2) {
        const int r = r0 + b2;
        if (16 * r < n_w) {
#pragma unroll
          for (int u = 0; u < VU; ++u) pv(16 * r + g + u * GP, vr[b2][u]);
          if (16 * (r + VD) < n_w) vload_run(r + VD, vr[b2]);
        }
      }
    }
  } else if (n_w > 0) {
#pragma unroll
    for (int u = 0; u < VU; ++u) pv(g + u * GP, vpre[u]);
    if (n_w > 16) {
#pragma unroll
      for (int u = 0; u < VU; ++u) pv(16 + g + u * GP, vpre2[u]);
    }
    for (int j0 = 32; j0 < n_w; j0 += 16) {  // later 16-position runs: run j0 + 16 is requested before run j0 is consumed
      kv8 vv[VU];
#pragma unroll
      for (int u = 0; u < VU; ++u) vv[u] = vnext[u];
      if (j0 + 16 < n_w) vload_run((j0 >> 4) + 1, vnext);
#pragma unroll
      for (int u = 0; u < VU; ++u) pv(j0 + g + u * GP, vv[u]);
    }
  }
  // sum over the GP row groups: lane groups of LPR lanes -> lanes 0 .. LPR - 1
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = acc[i];
    const uint32_t b32 = __float_as_uint(v);
    v += __uint_as_float(__builtin_amdgcn_permlane32_swap(b32, b32, false, false)[1]);
    const uint32_t b16 = __float_as_uint(v);
    v += __uint_as_float(__builtin_amdgcn_permlane16_swap(b16, b16, false, false)[1]);
    if constexpr (LPR == 8) v += WOQ_DPP_F32(v, 0x108);  // row_shl:8 — lane l += lane l + 8
    acc[i] = v;
  }
  if (lane < LPR) {
    float* dst = slab + wid * HD + l8 * 8;
    *(float4_t*)dst = float4_t{acc[0], acc[1], acc[2], acc[3]};
    *(float4_t*)(dst + 4) = float4_t{acc[4], acc[5], acc[6], acc[7]};
  }
  if (lane == 0) {
    ml[wid] = m_w;
    ml[4 + wid] = l_w;
  }
  if constexpr (SRC::granules) {
    if (has_new) new_position();  // wave 0: the launch's k / v strips are the last thing this workgroup waits for
  }
  env.sync();
  // ---- merge the four waves' partials ----
  if (tid < HD) {
    const float m0 = ml[0], m1 = ml[1], m2 = ml[2], m3 = ml[3];
    const float s_n = ml[8];  // the new position's score, -inf when it is not this workgroup's
    const float mx = fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), s_n);
    const float e_n = s_n == -INFINITY ? 0.f : __expf(s_n - mx);
    const float e0 = m0 == -INFINITY ? 0.f : __expf(m0 - mx), e1 = m1 == -INFINITY ? 0.f : __expf(m1 - mx);
    const float e2 = m2 == -INFINITY ? 0.f : __expf(m2 - mx), e3 = m3 == -INFINITY ? 0.f : __expf(m3 - mx);
    const float o4 = fmaf(slab[tid], e0, slab[HD + tid] * e1) + fmaf(slab[2 * HD + tid], e2, slab[3 * HD + tid] * e3);
    const float o = s_n == -INFINITY ? o4 : fmaf(vn[tid], e_n, o4);
    const float den = (fmaf(ml[4], e0, ml[5] * e1) + fmaf(ml[6], e2, ml[7] * e3)) + e_n;
    if constexpr (SPLIT && SRC::granules) {  // the slice's partial as tagged granules; merged below, by the slices themselves
      const AttnA2A a2a{src.part_g, src.tag, src.status};
      attn_a2a_publish(a2a, h, slice, HD, tid, o);
      if (tid == 0) {
        attn_a2a_publish(a2a, h, slice, HD, HD, mx);
        attn_a2a_publish(a2a, h, slice, HD, HD + 1, den);
      }
    } else if constexpr (SPLIT) {  // publish the slice's partial (woq_attn_merge.h); `out` is the partial buffer here
      st_agent(attn_part_o(out, h, slice, HD) + tid, o);
      if (tid == 0) {
        float* pml = attn_part_ml(out, heads, h, slice, HD);
        st_agent(pml, mx);
        st_agent(pml + 1, den);
      }
    } else {
      env.put(o / den, h * HD + tid);  // tid < HD is a whole number of 16-lane rows
    }
  }
  if constexpr (SPLIT && SRC::granules) {
    // ---- all-to-all merge: DPP row R of waves 0 / 1 finalises block b = slice + R * ns of this head ----
    const int R = wid * 4 + (lane >> 4), b = slice + R * n_slices;
    if (wid < 2 && slice < HD / 16)  // (wave-uniform: some row of this wave has a block)
      attn_a2a_finalize<HD, ATTN_A2A_MAX_SLICES>(AttnA2A{src.part_g, src.tag, src.status}, h, b, n_slices,
                                                 b < HD / 16, [&](float v, int idx) { env.put(v, idx); });
  }
}

template <typename KV, int HD, bool SPLIT, typename SRC>
__device__ __forceinline__ void attn_decode_body(float* sm, int h, int slice, int n_slices, const SRC& src,
                                                 KV* __restrict__ kcache, KV* __restrict__ vcache,
                                                 const int32_t* __restrict__ pos_p, const float* __restrict__ cs,
                                                 const float* __restrict__ sn, int heads, int kv_heads, int window,
                                                 int spw, float* __restrict__ out, const XqPtrs& xo,
                                                 const XqPub& pub = XqPub{nullptr, 0u}) {
  attn_decode_env<KV, HD, SPLIT>(sm, h, slice, n_slices, src, kcache, vcache, pos_p, cs, sn, heads, kv_heads, window,
                                 spw, out, AttnWgEnv{out, xo, pub});
}

}  // namespace woq
