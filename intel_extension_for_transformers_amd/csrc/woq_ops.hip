// woq_ops.hip — the ops between the quantised linears, as gfx950 kernels.
//
// In the reference these are stock HF transformers module forwards run by PyTorch CPU ops
// (LlamaRMSNorm, apply_rotary_pos_emb, SiLU*mul in LlamaMLP, GPT-2 gelu_new,
// eager attention over the KV cache); ITREX contributes no code there. All are HBM/latency-bound
// vector kernels: 16-byte loads, fp32 math, wave64 shuffles; no MFMA.
#include "woq_device.h"
#include "woq_launch.h"
#include "woq_xq.h"

namespace woq {

// ---- RMSNorm: one workgroup (256 threads) per row ------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const void* __restrict__ x, int dtype,
                                                      const float* __restrict__ w, float eps, int d,
                                                      void* __restrict__ out, int out_dtype) {
  __shared__ float part[4];
  const size_t row = blockIdx.x;
  const int tid = threadIdx.x;
  float ss = 0.f;
  for (int i = tid; i < d; i += 256) {
    float v = load_f32(x, row * d + i, dtype);
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  const float tot = part[0] + part[1] + part[2] + part[3];
  const float inv = 1.0f / sqrtf(tot / (float)d + eps);
  for (int i = tid; i < d; i += 256) {
    float v = load_f32(x, row * d + i, dtype);
    store_f32(out, row * d + i, out_dtype, v * inv * w[i]);
  }
}

// ---- RoPE (rotate_half), in place on [tokens, heads, D] -------------------------------------------
__global__ void rope_kernel(void* __restrict__ x, int dtype, const int32_t* __restrict__ pos,
                            const float* __restrict__ cs, const float* __restrict__ sn, int heads, int D,
                            size_t total_pairs) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_pairs) return;
  const int half = D >> 1;
  const int i = (int)(idx % half);
  const size_t th = idx / half;  // token*heads + head
  const int t = (int)(th / heads);
  const int p = pos[t];
  const float c = cs[(size_t)p * half + i], s = sn[(size_t)p * half + i];
  const size_t base = th * D;
  const float a = load_f32(x, base + i, dtype), b = load_f32(x, base + i + half, dtype);
  store_f32(x, base + i, dtype, a * c - b * s);
  store_f32(x, base + i + half, dtype, b * c + a * s);
}

__global__ void silu_mul_kernel(const void* __restrict__ g, const void* __restrict__ u, int dtype, size_t n,
                                void* __restrict__ out) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; idx < n; idx += stride) {
    const float a = load_f32(g, idx, dtype), b = load_f32(u, idx, dtype);
    store_f32(out, idx, dtype, a / (1.0f + expf(-a)) * b);
  }
}

__global__ void gelu_kernel(const void* __restrict__ x, int dtype, size_t n, int approximate,
                            void* __restrict__ out) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; idx < n; idx += stride) {
    const float v = load_f32(x, idx, dtype);
    float r;
    if (approximate)
      r = 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    else
      r = 0.5f * v * (1.0f + erff(v * 0.7071067811865476f));
    store_f32(out, idx, dtype, r);
  }
}

// ---- engine kernels (batch-1 decode) --------------------------------------------------------------

// hidden[h] = embed[token][h]  (fp32 residual stream)
// xo (optional): the row also leaves as the first layer's XQ vector (times its input norm weight) with the per-block
// sums of squares — hidden % 16 == 0, so every 16-lane row is one whole block
__global__ void embed_kernel(const void* __restrict__ embed, int dtype, const int32_t* __restrict__ token, int hidden,
                             float* __restrict__ out, const float* __restrict__ norm_w, XqPtrs xo,
                             float* __restrict__ ssq_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hidden) {
    const float v = load_f32(embed, (size_t)token[0] * hidden + i, dtype);
    out[i] = v;
    if (xo.limbs != nullptr) {
      const float ss = row16_sum(v * v);
      if ((i & 15) == 0) ssq_out[i >> 4] = ss;
      xq_emit16(v * norm_w[i], xo, i >> 4, i & 15);
    }
  }
}

// Single-query attention for one new token: one workgroup (4 waves) per query head.
//   qkv: fp32 [(heads + 2*kv_heads) * HD] un-rotated projections of the new token.
//   RoPE is applied here to q and to the new k; rotated k and v are appended to the cache (by the first query
//   head of each kv group) at position pos. kv caches: [max_ctx, kv_heads, HD] (fp16 | bf16 | e4m3). out fp32 [heads*HD].
// Three vectorised phases over the cached positions t < pos (the new position is taken from LDS, so no
// workgroup ever reads a cache row another workgroup is writing):
//   scores : 4 lanes per position (HD/4 dims each, 16-B loads), 16 positions per wave per iteration
//   softmax: block max / sum over the LDS score row
//   P.V    : HD/8 lanes per position (one 16-B load each), 4x unrolled, fp32 accumulate
// SPLIT (long contexts, flash-decoding style): gridDim.y workgroups per head, each over its own slice of the cached
// positions (the last slice also takes the new position) -> un-normalised partial (o[HD], max, sum) per (head,
// slice) in `out`; attn_combine_kernel merges them. A lone workgroup per head streams the cache at one CU's
// ~10 B/clk, which is why contexts beyond a few thousand positions need the slices.
template <typename KV, int HD, bool SPLIT>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ qkv, KV* __restrict__ kcache,
                                                          KV* __restrict__ vcache, const int32_t* __restrict__ pos_p,
                                                          const float* __restrict__ cs, const float* __restrict__ sn,
                                                          int heads, int kv_heads, int window,
                                                          float* __restrict__ out, XqPtrs xo) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  typedef typename KvVec8<KV>::type kv8;
  constexpr int half = HD / 2;
  constexpr int DPL = HD / 4;   // dims per lane in the score phase
  constexpr int LPR = HD / 8;   // lanes per row in the P.V phase
  constexpr int GP = 64 / LPR;  // position groups per wave in the P.V phase
  // workgroup ids go round-robin over the 8 XCDs: give every XCD a run of consecutive heads, so that the query heads
  // sharing a kv head (GQA) share an L2 instead of pulling the same cache rows into several
  const int bx = (int)blockIdx.x;
  const int h = (heads & 7) == 0 ? (bx & 7) * (heads >> 3) + (bx >> 3) : bx;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int rep = heads / kv_heads, kh = h / rep;
  const int apos = pos_p[0];  // absolute position of the new token = number of cached positions
  // sliding window (HF Mistral `sliding_window`, 0 = none): the query sees positions [apos + 1 - window, apos]
  const int w_lo = window > 0 ? max(0, apos + 1 - window) : 0;
  int t_lo = w_lo, pos = apos - w_lo;  // this workgroup's cached slice is [t_lo, t_lo + pos)
  bool incl_new = true;
  if constexpr (SPLIT) {
    const int ns = (int)gridDim.y, sp = (int)blockIdx.y;
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
      float* part = out + ((size_t)h * gridDim.y + blockIdx.y) * (HD + 2);
      part[tid] = o;
      if (tid == 0) {
        part[HD] = mx;
        part[HD + 1] = den;
      }
    } else {
      out[(size_t)h * HD + tid] = o / den;
      // the o_proj GEMV's XQ input: tid < HD is a whole number of 16-lane rows, one block each
      if (xo.limbs != nullptr) xq_emit16(o / den, xo, (h * HD + tid) >> 4, tid & 15);
    }
  }
}

// merge the slices of attn_decode_kernel<SPLIT>: out[h][d] = sum_s o_s[d] e^(m_s - m) / sum_s l_s e^(m_s - m).
// 256 threads per head: the slice maxima / sums go through LDS once, then two thread groups of HD walk alternate
// slices with their loads unrolled (independent addresses) — a serial per-thread walk over 32 slices of freshly
// written partials cost 14 us of pure load latency.
template <int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part, int ns,
                                                          float* __restrict__ out, XqPtrs xo) {
  __shared__ float ms[64], wl[64], wsc[64], red[256];
  const int h = blockIdx.x, tid = threadIdx.x;
  const float* p = part + (size_t)h * ns * (HD + 2);
  if (tid < ns) {
    ms[tid] = p[(size_t)tid * (HD + 2) + HD];
    wl[tid] = p[(size_t)tid * (HD + 2) + HD + 1];
  }
  __syncthreads();
  float m = -INFINITY;
  for (int s = 0; s < ns; ++s) m = fmaxf(m, ms[s]);
  if (tid < ns) {
    const float w = ms[tid] == -INFINITY ? 0.f : __expf(ms[tid] - m);
    wsc[tid] = w;
    wl[tid] *= w;
  }
  __syncthreads();
  float l = 0.f;
  for (int s = 0; s < ns; ++s) l += wl[s];
  constexpr int GROUPS = 256 / HD;  // 2 for head_dim 128, 4 for 64
  const int d = tid % HD, grp = tid / HD;
  float o = 0.f;
#pragma unroll 8
  for (int s = grp; s < ns; s += GROUPS) o = fmaf(p[(size_t)s * (HD + 2) + d], wsc[s], o);
  red[tid] = o;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < GROUPS; ++g2) t += red[g2 * HD + d];
    out[(size_t)h * HD + d] = t / l;
    if (xo.limbs != nullptr) xq_emit16(t / l, xo, (h * HD + d) >> 4, d & 15);
  }
}

// logits[v] = sum_h xn[h] * W[v][h], W dense fp16/bf16 [vocab, hidden] (lm_head is NOT quantised:
// utils/config.py:836-837). Final RMSNorm fused in the prologue. One wave per vocab row, 4 rows per WG.
__global__ __launch_bounds__(256) void lm_head_kernel(const float* __restrict__ hidden_in,
                                                      const float* __restrict__ norm_w, float eps,
                                                      const void* __restrict__ W, int w_dtype, int hidden, int vocab,
                                                      float* __restrict__ logits, float* __restrict__ pmax,
                                                      int32_t* __restrict__ pidx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;  // [hidden]
  __shared__ float part[4];
  __shared__ float bestv[4];
  __shared__ int besti[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float wbest = -INFINITY;  // this wave's best (logit, row) — rows ascend, so '>' keeps the lowest index on ties
  int wbi = 0x7fffffff;
  float ss = 0.f;
  for (int i = tid; i < hidden; i += 256) {
    const float v = hidden_in[i];
    xs[i] = v;
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  if (lane == 0) part[wid] = ss;
  __syncthreads();
  const float inv = 1.0f / sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)hidden + eps);
  for (int i = tid; i < hidden; i += 256) xs[i] = xs[i] * inv * norm_w[i];
  __syncthreads();
  const int rows_per_wg = 16;
  for (int r = wid; r < rows_per_wg; r += 4) {
    const int v = blockIdx.x * rows_per_wg + r;
    if (v >= vocab) break;
    const uint16_t* wr = (const uint16_t*)W + (size_t)v * hidden;
    float acc = 0.f;
    for (int k0 = lane * 8; k0 < hidden; k0 += 512) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 raw = __builtin_nontemporal_load((const u32x4*)(wr + k0));
      const uint32_t rr[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo, hi;
        if (w_dtype == WOQ_BF16) {
          lo = bf16_bits_to_f32(rr[j] & 0xffff);
          hi = bf16_bits_to_f32(rr[j] >> 16);
        } else {
          lo = f16_bits_to_f32(rr[j] & 0xffff);
          hi = f16_bits_to_f32(rr[j] >> 16);
        }
        acc = fmaf(lo, xs[k0 + 2 * j], acc);
        acc = fmaf(hi, xs[k0 + 2 * j + 1], acc);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) logits[v] = acc;
    if (acc > wbest) {
      wbest = acc;
      wbi = v;
    }
  }
  // per-workgroup (max, index) for the greedy argmax: it then reduces vocab / 16 pairs instead of vocab logits
  if (pmax) {
    if (lane == 0) {
      bestv[wid] = wbest;
      besti[wid] = wbi;
    }
    __syncthreads();
    if (tid == 0) {
      float b = bestv[0];
      int bi = besti[0];
      for (int w = 1; w < 4; ++w)
        if (bestv[w] > b || (bestv[w] == b && besti[w] < bi)) {
          b = bestv[w];
          bi = besti[w];
        }
      pmax[blockIdx.x] = b;
      pidx[blockIdx.x] = bi;
    }
  }
}

// greedy: token = argmax(logits) (lowest index on ties, like torch.argmax), pos += 1. One workgroup.
// `cand` null: candidate i is logit i; else (logits[i], cand[i]) are the per-workgroup pairs lm_head_kernel left.
// `log` (optional): log[position of the token that was just fed] = the new token, so a host that replays several steps
// back to back can read them all afterwards instead of synchronising on every step.
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, const int32_t* __restrict__ cand,
                                                      int vocab, int32_t* __restrict__ token,
                                                      int32_t* __restrict__ pos, int32_t* __restrict__ log) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = tid; i < vocab; i += 1024) {
    const float v = logits[i];
    const int ci = cand ? cand[i] : i;
    if (v > best || (v == best && ci < idx)) {
      best = v;
      idx = ci;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) {
      best = ov;
      idx = oi;
    }
  }
  if (lane == 0) {
    bv[wid] = best;
    bi[wid] = idx;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) {
        best = bv[w];
        idx = bi[w];
      }
    token[0] = idx;
    if (log != nullptr) log[pos[0]] = idx;
    pos[0] = pos[0] + 1;
  }
}

// ---- host launchers used by the engine ------------------------------------------------------------
void launch_embed(const void* embed, int dtype, const int32_t* token, int hidden, float* out, const float* norm_w,
                  const XqPtrs& xo, float* ssq_out, hipStream_t st) {
  hipLaunchKernelGGL(embed_kernel, dim3((hidden + 255) / 256), dim3(256), 0, st, embed, dtype, token, hidden, out,
                     norm_w, xo, ssq_out);
}

bool launch_attn_decode_mfma(const float* qkv, void* kcache, void* vcache, int kv_dtype, const int32_t* pos,
                             const float* cs, const float* sn, int heads, int kv_heads, int D, int window, int splits,
                             float* part, hipStream_t st);
void launch_attn_combine(const float* part, int heads, int D, int splits, float* out, const XqPtrs& xo,
                         hipStream_t st);

template <typename KV, int HD>
static int launch_attn_t(const float* qkv, void* kcache, void* vcache, const int32_t* pos, const float* cs,
                         const float* sn, int heads, int kv_heads, int max_ctx, int window, float* out, int splits,
                         float* part, const XqPtrs& xo, hipStream_t st) {
  constexpr int GP = 64 / (HD / 8);
  const int reach = window > 0 ? min(window, max_ctx) : max_ctx;  // positions a query can see
  const int span = splits > 1 ? ((((reach + splits - 1) / splits) + 63) & ~63) + 64 : reach;
  const size_t lds = (size_t)(3 * HD + 8 + 4 * GP * HD + ((span + 4) & ~3)) * 4;
  if (lds > 160 * 1024) return woq::fail("QBits: max_ctx too large for the decode attention (raise attn_splits)");
  if (splits > 1) {
    auto k = attn_decode_kernel<KV, HD, true>;
    static bool once = false;
    if (!once) {
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      once = true;
    }
    hipLaunchKernelGGL(k, dim3(heads, splits), dim3(256), lds, st, qkv, (KV*)kcache, (KV*)vcache, pos, cs, sn, heads,
                       kv_heads, window, part, XqPtrs{nullptr, nullptr, nullptr});
    hipLaunchKernelGGL(attn_combine_kernel<HD>, dim3(heads), dim3(256), 0, st, part, splits, out, xo);
    return 0;
  }
  auto k = attn_decode_kernel<KV, HD, false>;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipLaunchKernelGGL(k, dim3(heads), dim3(256), lds, st, qkv, (KV*)kcache, (KV*)vcache, pos, cs, sn, heads, kv_heads,
                     window, out, xo);
  return 0;
}

// splits <= 1: one workgroup per head (short contexts); else `splits` slices per head + a combine launch, partials in
// `part` (fp32 [heads][splits][D + 2]).
int launch_attn_decode(const float* qkv, void* kcache, void* vcache, int kv_dtype, const int32_t* pos,
                       const float* cs, const float* sn, int heads, int kv_heads, int D, int max_ctx, int window,
                       float* out, int splits, int grouped, float* part, const XqPtrs& xo, hipStream_t st) {
  if (D != 64 && D != 128) return woq::fail("QBits: attention head_dim must be 64 or 128");
  // grouped-query form (woq_prefill.hip): only where it applies — head_dim 128, 2 / 4 / 8 query heads per kv head,
  // an fp16 or fp8 cache — anything else keeps the per-query-head slices
  if (grouped && launch_attn_decode_mfma(qkv, kcache, vcache, kv_dtype, pos, cs, sn, heads, kv_heads, D, window, splits,
                                          part, st)) {
    launch_attn_combine(part, heads, D, splits, out, xo, st);
    return 0;
  }
#define WOQ_ATTN_DEC(T)                                                                                              \
  return D == 128 ? launch_attn_t<T, 128>(qkv, kcache, vcache, pos, cs, sn, heads, kv_heads, max_ctx, window, out,   \
                                          splits, part, xo, st)                                                     \
                  : launch_attn_t<T, 64>(qkv, kcache, vcache, pos, cs, sn, heads, kv_heads, max_ctx, window, out,    \
                                         splits, part, xo, st);
  if (kv_dtype == WOQ_F16) { WOQ_ATTN_DEC(_Float16) }
  if (kv_dtype == WOQ_FP8_E4M3) { WOQ_ATTN_DEC(Fp8) }
  WOQ_ATTN_DEC(__bf16)
#undef WOQ_ATTN_DEC
}

void launch_attn_combine(const float* part, int heads, int D, int splits, float* out, const XqPtrs& xo,
                         hipStream_t st) {
  if (D == 128)
    hipLaunchKernelGGL(attn_combine_kernel<128>, dim3(heads), dim3(256), 0, st, part, splits, out, xo);
  else
    hipLaunchKernelGGL(attn_combine_kernel<64>, dim3(heads), dim3(256), 0, st, part, splits, out, xo);
}

// pmax / pidx (nullable): per-workgroup (max logit, its index), (vocab + 15) / 16 entries each
void launch_lm_head(const float* hidden_in, const float* norm_w, float eps, const void* W, int w_dtype, int hidden,
                    int vocab, float* logits, float* pmax, int32_t* pidx, hipStream_t st) {
  hipLaunchKernelGGL(lm_head_kernel, dim3((vocab + 15) / 16), dim3(256), (size_t)hidden * 4, st, hidden_in, norm_w,
                     eps, W, w_dtype, hidden, vocab, logits, pmax, pidx);
}

void launch_argmax(const float* logits, int vocab, int32_t* token, int32_t* pos, hipStream_t st) {
  hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, logits, (const int32_t*)nullptr, vocab, token, pos,
                     (int32_t*)nullptr);
}

// greedy token from the (max, index) pairs of launch_lm_head
void launch_argmax_pairs(const float* pmax, const int32_t* pidx, int n, int32_t* token, int32_t* pos, int32_t* log,
                         hipStream_t st) {
  hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, pmax, pidx, n, token, pos, log);
}

}  // namespace woq

using namespace woq;

extern "C" {

int woq_rmsnorm(const void* x_dev, int dtype, const float* weight_dev, float eps, int rows, int d, void* out_dev,
                int out_dtype, void* stream) {
  WOQ_TRY
  WOQ_CHECK(rows >= 0 && d > 0, "QBits: bad rmsnorm shape");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x_dev, dtype, weight_dev, eps, d,
                     out_dev, out_dtype);
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

int woq_rope(void* x_dev, int dtype, const int32_t* pos_dev, const float* cos_dev, const float* sin_dev, int tokens,
             int heads, int D, void* stream) {
  WOQ_TRY
  WOQ_CHECK((D & 1) == 0, "QBits: rope head_dim must be even");
  size_t total = (size_t)tokens * heads * (D / 2);
  if (total == 0) return 0;
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev,
                     dtype, pos_dev, cos_dev, sin_dev, heads, D, total);
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

int woq_silu_mul(const void* gate_dev, const void* up_dev, int dtype, size_t n, void* out_dev, void* stream) {
  WOQ_TRY
  if (n == 0) return 0;
  unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(silu_mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gate_dev, up_dev, dtype, n,
                     out_dev);
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

int woq_gelu(const void* x_dev, int dtype, size_t n, int approximate, void* out_dev, void* stream) {
  WOQ_TRY
  if (n == 0) return 0;
  unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(gelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_dev, dtype, n, approximate,
                     out_dev);
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

}  // extern "C"
