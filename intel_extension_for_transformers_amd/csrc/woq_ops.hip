// woq_ops.hip — the ops between the quantised linears, as gfx950 kernels.
//
// In the reference these are stock HF transformers module forwards run by PyTorch CPU ops
// (LlamaRMSNorm, apply_rotary_pos_emb, SiLU*mul in LlamaMLP, GPT-2 gelu_new,
// eager attention over the KV cache); ITREX contributes no code there. All are HBM/latency-bound
// vector kernels: 16-byte loads, fp32 math, wave64 shuffles; no MFMA.
#include "woq_attn_decode.h"
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
                             float* __restrict__ ssq_out, unsigned int* __restrict__ step_seq,
                             int32_t* __restrict__ pos, int max_ctx, int* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {  // first kernel of a decode step
    if (step_seq != nullptr) step_seq[0] = step_seq[0] + 1u;  // new hand-off tags
    // the position lives on the device (greedy chaining advances it there): the KV cache, the RoPE tables and the token
    // log are indexed by it unchecked, so a step that starts out of range is moved to the last slot and flagged
    if (pos != nullptr && pos[0] >= max_ctx) {
      pos[0] = max_ctx - 1;
      if (status != nullptr) atomicOr(status, 2);
    }
  }
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

// The greedy tail of step t and the head of step t + 1 in ONE launch (round 5; inside a graph of chained steps every
// interior token boundary was argmax [1 workgroup] -> boundary -> embed [hidden / 256 workgroups]): every workgroup
// reduces the lm_head's (max, index) pairs itself — vocab / 16 pairs, 8 per thread, the same tie rule (lowest index) in
// the same order everywhere, so all agree — and embeds its 256 columns of the winner's row; workgroup 0 also does the
// bookkeeping of both kernels (token, token log, position + 1, the new step's hand-off tag, the max_ctx guard).
__global__ __launch_bounds__(256) void argmax_embed_kernel(const float* __restrict__ pmax, const int32_t* __restrict__ pidx,
                                                           int n_pairs, int32_t* __restrict__ token,
                                                           int32_t* __restrict__ pos, int32_t* __restrict__ log,
                                                           const void* __restrict__ embed, int dtype, int hidden,
                                                           float* __restrict__ out, const float* __restrict__ norm_w,
                                                           XqPtrs xo, float* __restrict__ ssq_out,
                                                           unsigned int* __restrict__ step_seq, int max_ctx,
                                                           int* __restrict__ status) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = tid; i < n_pairs; i += 256) {
    const float v = pmax[i];
    const int ci = pidx[i];
    if (v > best || (v == best && ci < idx)) best = v, idx = ci;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) best = ov, idx = oi;
  }
  if (lane == 0) bv[wid] = best, bi[wid] = idx;
  __syncthreads();
  best = bv[0], idx = bi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (bv[w] > best || (bv[w] == best && bi[w] < idx)) best = bv[w], idx = bi[w];
  // no candidate won (every logit NaN): the sentinel must not index the embedding table — token 0 and a status bit
  const bool no_winner = idx < 0 || (size_t)idx >= (size_t)n_pairs * 16;
  if (no_winner) idx = 0;
  if (blockIdx.x == 0 && tid == 0) {
    if (no_winner && status != nullptr) atomicOr(status, 4);
    token[0] = idx;
    int p = pos[0];
    if (log != nullptr) log[p] = idx;
    p += 1;
    if (step_seq != nullptr) step_seq[0] = step_seq[0] + 1u;
    if (p >= max_ctx) {  // the next step would start out of range: the embedding kernel's guard
      p = max_ctx - 1;
      if (status != nullptr) atomicOr(status, 2);
    }
    pos[0] = p;
  }
  const int i = blockIdx.x * 256 + tid;
  if (i < hidden) {
    const float v = load_f32(embed, (size_t)idx * hidden + i, dtype);
    out[i] = v;
    if (xo.limbs != nullptr) {
      const float ss = row16_sum(v * v);
      if ((i & 15) == 0) ssq_out[i >> 4] = ss;
      xq_emit16(v * norm_w[i], xo, i >> 4, i & 15);
    }
  }
}

template <typename KV, int HD, bool SPLIT>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ qkv, KV* __restrict__ kcache,
                                                          KV* __restrict__ vcache, const int32_t* __restrict__ pos_p,
                                                          const float* __restrict__ cs, const float* __restrict__ sn,
                                                          int hk, int window, int spw,
                                                          float* __restrict__ out, XqPtrs xo, AttnMerge mg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // hk = heads | kv_heads << 16: with `window` the 13th and 14th argument dwords — everything in front of the first
  // K / V request is preloaded (the argument segment's first read costs ~1 us, profiles/r06c_xqs_stage_stamps.txt)
  const int heads = hk & 0xffff, kv_heads = hk >> 16;
  // workgroup ids go round-robin over the 8 XCDs: give every XCD a run of consecutive heads, so that the query heads
  // sharing a kv head (GQA) share an L2 instead of pulling the same cache rows into several
  const int bx = (int)blockIdx.x;
  const int h = (heads & 7) == 0 ? (bx & 7) * (heads >> 3) + (bx >> 3) : bx;
  attn_decode_body<KV, HD, SPLIT>(sm, h, (int)blockIdx.y, (int)gridDim.y, AttnPlain{qkv}, kcache, vcache, pos_p, cs,
                                  sn, heads, kv_heads, window, spw, out, xo);
  if constexpr (SPLIT) {  // `out` = the partial buffer; the head's last slice workgroup merges (woq_attn_merge.h)
    if (mg.counter != nullptr) attn_slices_merge<HD, 1>(out, heads, h, 1, (int)gridDim.y, mg.counter + h, mg, sm);
  }
}

// merge the slices of attn_decode_kernel<SPLIT>: out[h][d] = sum_s o_s[d] e^(m_s - m) / sum_s l_s e^(m_s - m).
// 256 threads per head; two thread groups of HD walk alternate slices. Round 4: ONE round trip — a thread's partial
// rows (clamped, branch-free; rows past ns weigh 0) are requested together with the slice maxima / sums, before anything
// is waited for; the weights are then worked out in LDS while the rows are in flight (round 3 read the maxima, then
// the rows: two dependent trips in a launch that is nothing but latency, 5.1 us per layer at 32 slices).
template <int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part, int ns,
                                                          float* __restrict__ out, XqPtrs xo) {
  __shared__ float ms[64], wl[64], wsc[64], red[256];
  constexpr int GROUPS = 256 / HD;                // 2 for head_dim 128, 4 for 64
  constexpr int PER = ATTN_MAX_SLICES / GROUPS;   // partial rows a thread may have to fetch
  const int h = blockIdx.x, tid = threadIdx.x;
  const int d = tid % HD, grp = tid / HD;
  const float* p = part + (size_t)h * ATTN_MAX_SLICES * HD;  // woq_attn_merge.h: o [head][64][HD], then ml [head][64][2]
  const float* pml = part + (size_t)gridDim.x * ATTN_MAX_SLICES * HD + (size_t)h * ATTN_MAX_SLICES * 2;
  const int si = min(tid, ns - 1);
  const float m_r = pml[si * 2], l_r = pml[si * 2 + 1];
  float v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) v[u] = p[(size_t)min(grp + u * GROUPS, ns - 1) * HD + d];
  if (tid < ns) ms[tid] = m_r, wl[tid] = l_r;
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
  float o = 0.f;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int s = grp + u * GROUPS;
    if (s < ns) o = fmaf(v[u], wsc[s], o);  // ascending s
  }
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
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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
  auto consume = [&](const u32x4& raw, int k0, float& acc) {
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
  };
  auto finish = [&](int v, float acc) {
    acc = wave_sum(acc);
    if (lane == 0) logits[v] = acc;
    if (acc > wbest) {
      wbest = acc;
      wbi = v;
    }
  };
  for (int r = wid; r < rows_per_wg; r += 4) {
    const int v = blockIdx.x * rows_per_wg + r;
    if (v >= vocab) break;
    const uint16_t* wr = (const uint16_t*)W + (size_t)v * hidden;
    float acc = 0.f;
    for (int k0 = lane * 8; k0 < hidden; k0 += 512) consume(__builtin_nontemporal_load((const u32x4*)(wr + k0)), k0, acc);
    finish(v, acc);
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
                  const XqPtrs& xo, float* ssq_out, unsigned int* step_seq, int32_t* pos, int max_ctx, int* status,
                  hipStream_t st) {
  hipLaunchKernelGGL(embed_kernel, dim3((hidden + 255) / 256), dim3(256), 0, st, embed, dtype, token, hidden, out,
                     norm_w, xo, ssq_out, step_seq, pos, max_ctx, status);
}

bool launch_attn_decode_mfma(const float* qkv, void* kcache, void* vcache, int kv_dtype, const int32_t* pos,
                             const float* cs, const float* sn, int heads, int kv_heads, int D, int window, int splits,
                             float* part, int chunk_fixed, int max_ctx, const AttnMerge& mg, hipStream_t st);
void launch_attn_combine(const float* part, int heads, int D, int splits, float* out, const XqPtrs& xo,
                         hipStream_t st);

template <typename KV, int HD>
static int launch_attn_t(const float* qkv, void* kcache, void* vcache, const int32_t* pos, const float* cs,
                         const float* sn, int heads, int kv_heads, int max_ctx, int window, float* out, int splits,
                         float* part, const XqPtrs& xo, unsigned int* merge_counters, hipStream_t st) {
  const int reach = window > 0 ? min(window, max_ctx) : max_ctx;  // positions a query can see
  const int span = splits > 1 ? ((((reach + splits - 1) / splits) + 63) & ~63) + 64 : reach;
  const int spw = attn_dec_spw(span);
  const size_t lds = attn_dec_lds_floats(HD, span) * 4;
  if (lds > 160 * 1024) return woq::fail("QBits: max_ctx too large for the decode attention (raise attn_splits)");
  if (splits > 1) {
    auto k = attn_decode_kernel<KV, HD, true>;
    static bool once = false;
    if (!once) {
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      once = true;
    }
    const AttnMerge mg{merge_counters, out, xo};
    hipLaunchKernelGGL(k, dim3(heads, splits), dim3(256), lds, st, qkv, (KV*)kcache, (KV*)vcache, pos, cs, sn,
                       heads | (kv_heads << 16), window, spw, part, XqPtrs{nullptr, nullptr, nullptr}, mg);
    if (merge_counters == nullptr)
      hipLaunchKernelGGL(attn_combine_kernel<HD>, dim3(heads), dim3(256), 0, st, part, splits, out, xo);
    return 0;
  }
  auto k = attn_decode_kernel<KV, HD, false>;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipLaunchKernelGGL(k, dim3(heads), dim3(256), lds, st, qkv, (KV*)kcache, (KV*)vcache, pos, cs, sn,
                     heads | (kv_heads << 16), window, spw, out, xo,
                     AttnMerge{nullptr, nullptr, XqPtrs{nullptr, nullptr, nullptr}});
  return 0;
}

// splits <= 1: one workgroup per head (short contexts); else `splits` slices per head + a combine launch, partials in
// `part` (fp32 [heads][splits][D + 2]).
// merge_counters (nullable): [heads] zero-initialised words — the slices' last workgroup merges (woq_attn_merge.h) and
// no combine launch follows; chunk_fixed: position-independent slice geometry of the grouped form (0 = adaptive)
int launch_attn_decode(const float* qkv, void* kcache, void* vcache, int kv_dtype, const int32_t* pos,
                       const float* cs, const float* sn, int heads, int kv_heads, int D, int max_ctx, int window,
                       float* out, int splits, int grouped, float* part, const XqPtrs& xo, hipStream_t st,
                       unsigned int* merge_counters, int chunk_fixed, const AttnA2A* a2a_grouped, const unsigned int* seq,
                       int layer) {
  if (D != 64 && D != 128) return woq::fail("QBits: attention head_dim must be 64 or 128");
  if (splits > ATTN_MAX_SLICES) return woq::fail("QBits: at most 64 context slices");
  // grouped-query form (woq_prefill.hip): only where it applies — head_dim 128, 2 / 4 / 8 query heads per kv head,
  // an fp16 or fp8 cache — anything else keeps the per-query-head slices
  // a2a_grouped (round 6, nullable; grouped form only): the slices merge among themselves through tagged granules
  // (woq_attn_merge.h) — no combine launch; the caller has checked that the grid can be resident at once
  AttnMerge mg{merge_counters, out, xo};
  const bool a2a = grouped && a2a_grouped != nullptr && merge_counters == nullptr && splits <= 32;
  if (a2a) mg.part_g = a2a_grouped->part_g, mg.seq = seq, mg.layer = layer, mg.status = a2a_grouped->status;
  if (grouped && launch_attn_decode_mfma(qkv, kcache, vcache, kv_dtype, pos, cs, sn, heads, kv_heads, D, window, splits,
                                          part, chunk_fixed, max_ctx, mg, st)) {
    if (merge_counters == nullptr && !a2a) launch_attn_combine(part, heads, D, splits, out, xo, st);
    return 0;
  }
#define WOQ_ATTN_DEC(T)                                                                                              \
  return D == 128 ? launch_attn_t<T, 128>(qkv, kcache, vcache, pos, cs, sn, heads, kv_heads, max_ctx, window, out,   \
                                          splits, part, xo, merge_counters, st)                                     \
                  : launch_attn_t<T, 64>(qkv, kcache, vcache, pos, cs, sn, heads, kv_heads, max_ctx, window, out,    \
                                         splits, part, xo, merge_counters, st);
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

// greedy token of the step that just ran its lm_head + embedding row of the NEXT step, one launch (argmax_embed_kernel)
void launch_argmax_embed(const float* pmax, const int32_t* pidx, int n, int32_t* token, int32_t* pos, int32_t* log,
                         const void* embed, int dtype, int hidden, float* out, const float* norm_w, const XqPtrs& xo,
                         float* ssq_out, unsigned int* step_seq, int max_ctx, int* status, hipStream_t st) {
  hipLaunchKernelGGL(argmax_embed_kernel, dim3((hidden + 255) / 256), dim3(256), 0, st, pmax, pidx, n, token, pos, log,
                     embed, dtype, hidden, out, norm_w, xo, ssq_out, step_seq, max_ctx, status);
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
