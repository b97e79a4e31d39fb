// woq_attn_body.h — the batch-1 decode attention of one (head, slice) workgroup as a device function, shared by
// the stand-alone launch (woq_ops.hip attn_decode_kernel) and the launch that also carries the o_proj GEMV
// (woq_attn_o.hip). See woq_ops.hip for the description of the three phases.
#pragma once
#include "woq_device.h"
#include "woq_xq.h"

namespace woq {

// bx / by / ny: the workgroup's head index, slice index and slice count (the x / y block index and the y grid size
// of the stand-alone launch); sm: its dynamic LDS. XQ_SC1: the XQ output is written with agent-scope (sc1,
// write-through) stores — the form a consumer in the SAME launch needs (woq_attn_o.hip).
template <typename KV, int HD, bool SPLIT, bool XQ_SC1 = false>
__device__ __forceinline__ void attn_decode_body(const float* __restrict__ qkv, KV* __restrict__ kcache,
                                                 KV* __restrict__ vcache, const int32_t* __restrict__ pos_p,
                                                 const float* __restrict__ cs, const float* __restrict__ sn, int heads,
                                                 int kv_heads, int window, float* __restrict__ out, XqPtrs xo, int bx,
                                                 int by, int ny, float* sm) {
  typedef typename KvVec8<KV>::type kv8;
  constexpr int half = HD / 2;
  constexpr int DPL = HD / 4;   // dims per lane in the score phase
  constexpr int LPR = HD / 8;   // lanes per row in the P.V phase
  constexpr int GP = 64 / LPR;  // position groups per wave in the P.V phase
  // workgroup ids go round-robin over the 8 XCDs: give every XCD a run of consecutive heads, so that the query heads
  // sharing a kv head (GQA) share an L2 instead of pulling the same cache rows into several
  const int h = (heads & 7) == 0 ? (bx & 7) * (heads >> 3) + (bx >> 3) : bx;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int rep = heads / kv_heads, kh = h / rep;
  const int apos = pos_p[0];  // absolute position of the new token = number of cached positions
  // sliding window (HF Mistral `sliding_window`, 0 = none): the query sees positions [apos + 1 - window, apos]
  const int w_lo = window > 0 ? max(0, apos + 1 - window) : 0;
  int t_lo = w_lo, pos = apos - w_lo;  // this workgroup's cached slice is [t_lo, t_lo + pos)
  bool incl_new = true;
  if constexpr (SPLIT) {
    const int ns = ny, sp = by;
    const int span = apos - w_lo;
    const int chunk = (((span + ns - 1) / ns) + 63) & ~63;
    t_lo = w_lo + min(sp * chunk, span);
    pos = min(apos - t_lo, chunk);
    incl_new = sp == ns - 1;
  }
  kcache += (size_t)t_lo * kv_heads * HD;
  vcache += (size_t)t_lo * kv_heads * HD;
  const int npos_abs = apos - t_lo;  // row of the new position relative to the re-based cache pointers
  float* qs = sm;                 // [HD] rotated q (pre-scaled by 1/sqrt(HD))
  float* kn = qs + HD;            // [HD] rotated new k, rounded to the cache dtype
  float* vn = kn + HD;            // [HD] new v, rounded to the cache dtype
  float* redm = vn + HD;          // [8]
  float* slab = redm + 8;         // [4 waves][GP][HD] partial outputs
  float* sc = slab + 4 * GP * HD; // [pos + 1] scores / probabilities
  const float scale = 1.0f / sqrtf((float)HD);
  // The cache rows of the first score / P.V iteration depend only on `pos`: fetch them now, so that their HBM
  // latency runs under the q/k/v read, the RoPE and the first barrier instead of after them.
  const int sub = lane & 3;
  const int g = lane / LPR, l8 = lane % LPR;
  constexpr int TSTEP = 4 * GP;  // positions covered by the workgroup per P.V pass
  const int plast = max(pos - 1, 0);
  kv8 kpre[DPL / 8];
  {
    const int tc = min(wid * 16 + (lane >> 2), plast);
    const kv8* kp = (const kv8*)(kcache + ((size_t)tc * kv_heads + kh) * HD + sub * DPL);
#pragma unroll
    for (int j = 0; j < DPL / 8; ++j) kpre[j] = kp[j];
  }
  kv8 vpre[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int tc = min(wid * GP + g + u * TSTEP, plast);
    vpre[u] = *(const kv8*)(vcache + ((size_t)tc * kv_heads + kh) * HD + l8 * 8);
  }
  if (tid < half) {
    const float c = cs[(size_t)apos * half + tid], s = sn[(size_t)apos * half + tid];
    const float* q = qkv + (size_t)h * HD;
    const float* k = qkv + (size_t)(heads + kh) * HD;
    const float qa = q[tid], qb = q[tid + half], ka = k[tid], kb = k[tid + half];
    qs[tid] = (qa * c - qb * s) * scale;
    qs[tid + half] = (qb * c + qa * s) * scale;
    kn[tid] = (float)(KV)(ka * c - kb * s);
    kn[tid + half] = (float)(KV)(kb * c + ka * s);
  } else if (tid >= 128 && tid < 128 + HD) {
    vn[tid - 128] = (float)(KV)qkv[(size_t)(heads + kv_heads + kh) * HD + (tid - 128)];
  }
  __syncthreads();
  if (h % rep == 0 && tid < HD && incl_new) {
    kcache[((size_t)npos_abs * kv_heads + kh) * HD + tid] = (KV)kn[tid];
    vcache[((size_t)npos_abs * kv_heads + kh) * HD + tid] = (KV)vn[tid];
  }
  // ---- scores for cached positions ----
  float qreg[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) qreg[i] = qs[sub * DPL + i];
  float lmax = -INFINITY;
  auto score = [&](int t0, const kv8 (&kv)[DPL / 8]) {
    const int t = t0 + (lane >> 2);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < DPL / 8; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) d = fmaf(qreg[j * 8 + i], (float)kv[j][i], d);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    if (t < pos) {
      if (sub == 0) sc[t] = d;
      lmax = fmaxf(lmax, d);
    }
  };
  if (wid * 16 < pos) score(wid * 16, kpre);
  {
    // rows of iteration i+1 are in flight while iteration i is scored (a lone dependent load per iteration would
    // expose the full HBM latency every 64 positions)
    auto kload = [&](int t0, kv8 (&kv)[DPL / 8]) {
      const int tc = min(t0 + (lane >> 2), max(pos - 1, 0));
      const kv8* kp = (const kv8*)(kcache + ((size_t)tc * kv_heads + kh) * HD + sub * DPL);
#pragma unroll
      for (int j = 0; j < DPL / 8; ++j) kv[j] = kp[j];
    };
    kv8 ka[DPL / 8], kb[DPL / 8];
    int t0 = wid * 16 + 64;
    if (t0 < pos) kload(t0, ka);
    for (; t0 < pos; t0 += 128) {
      if (t0 + 64 < pos) kload(t0 + 64, kb);
      score(t0, ka);
      if (t0 + 64 < pos) {
        if (t0 + 128 < pos) kload(t0 + 128, ka);
        score(t0 + 64, kb);
      }
    }
  }
  if (tid == 0 && incl_new) {  // the new position, from LDS
    float d = 0.f;
    for (int i = 0; i < HD; ++i) d = fmaf(qs[i], kn[i], d);
    sc[pos] = d;
    lmax = fmaxf(lmax, d);
  }
  lmax = wave_max(lmax);
  if (lane == 0) redm[wid] = lmax;
  __syncthreads();
  const float mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  float lsum = 0.f;
  for (int t = tid; t < pos + (incl_new ? 1 : 0); t += 256) {
    const float p = __expf(sc[t] - mx);
    sc[t] = p;
    lsum += p;
  }
  lsum = wave_sum(lsum);
  if (lane == 0) redm[4 + wid] = lsum;
  __syncthreads();
  const float den = (redm[4] + redm[5]) + (redm[6] + redm[7]);
  // ---- P.V ----
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  auto pv = [&](int t0, const kv8 (&vv)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * TSTEP;
      const float p = t < pos ? sc[min(t, pos - 1)] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(p, (float)vv[u][i], acc[i]);
    }
  };
  if (wid * GP + g < pos) pv(wid * GP + g, vpre);
  {
    auto vload = [&](int t0, kv8 (&vv)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tc = min(t0 + u * TSTEP, max(pos - 1, 0));
        vv[u] = *(const kv8*)(vcache + ((size_t)tc * kv_heads + kh) * HD + l8 * 8);
      }
    };
    kv8 va[4], vb[4];
    int t0 = wid * GP + g + 4 * TSTEP;
    if (t0 < pos) vload(t0, va);
    for (; t0 < pos; t0 += 8 * TSTEP) {
      if (t0 + 4 * TSTEP < pos) vload(t0 + 4 * TSTEP, vb);
      pv(t0, va);
      if (t0 + 4 * TSTEP < pos) {
        if (t0 + 8 * TSTEP < pos) vload(t0 + 8 * TSTEP, va);
        pv(t0 + 4 * TSTEP, vb);
      }
    }
  }
  if (wid == 0 && g == 0 && incl_new) {  // the new position
    const float p = sc[pos];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(p, vn[l8 * 8 + i], acc[i]);
  }
  float* dst = slab + ((size_t)(wid * GP + g)) * HD + l8 * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = acc[i];
  __syncthreads();
  if (tid < HD) {
    float o = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 4 * GP; ++s2) o += slab[s2 * HD + tid];
    if constexpr (SPLIT) {
      float* part = out + ((size_t)h * ny + by) * (HD + 2);
      part[tid] = o;
      if (tid == 0) {
        part[HD] = mx;
        part[HD + 1] = den;
      }
    } else {
      out[(size_t)h * HD + tid] = o / den;
      // the o_proj GEMV's XQ input: tid < HD is a whole number of 16-lane rows, one block each
      if (xo.limbs != nullptr) xq_emit16<XQ_SC1>(o / den, xo, (h * HD + tid) >> 4, tid & 15);
    }
  }
}


}  // namespace woq
