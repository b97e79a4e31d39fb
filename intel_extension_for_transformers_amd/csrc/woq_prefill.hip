// woq_prefill.hip — the non-linear kernels of the prompt (prefill) pass of the decode engine: token embedding
// for a batch of rows, RoPE + KV-cache append, causal attention over the cache on the matrix cores, last-row gather.
//
// What they replace: stock HF module forwards the reference runs as PyTorch CPU ops around its
// qbits linears during `model.generate`'s first forward — LlamaRotaryEmbedding / apply_rotary_pos_emb (rotate_half
// form), the KV cache `torch.cat`, and LlamaAttention's softmax(QK^T / sqrt(d) + causal mask) V. The linears of the
// same pass are woq_gemm_f16.hip.
//
// Attention (attn_prefill_kernel): one workgroup = 128 query rows of one head of one sequence, 4 waves x 32 rows;
// the K / V rows of the sequence's cache — positions [0, start + row], i.e. earlier chunks and this chunk alike —
// stream through LDS in tiles of 64 positions:
//  * S^T = K Q^T per 16-position tile (v_mfma_f32_16x16x32_f16, A = K fragment read from a row-major LDS tile in a
//    16-B-slot XOR swizzle, B = Q^T fragments held in registers): the result leaves query i16 in lane i16 with
//    positions 4*kq + j — which IS the B-operand shape of the next product, so probabilities never touch LDS;
//  * online softmax in the exp2 domain, fp32, per query = per lane column (two cross-quarter shuffles per tile);
//  * O^T = V^T P^T: A = V^T fragment = two ds_read_b64 from a TRANSPOSED V tile ([d][position], written with a
//    4 x 8 register transpose and a quad-chunk XOR swizzle so both the writes and the reads are conflict-free),
//    B = the packed fp16 probabilities; the output again has query i16 in lane i16, so the running rescale is a
//    plain per-lane multiply.
// Tiles entirely above a wave's causal diagonal are skipped by that wave; the workgroups with the most tiles are
// scheduled first. One LDS buffer, two barriers per tile (a two-buffer, one-barrier variant measured 28 % slower:
// 450 vs 353 us per layer at 4 x 2048 tokens — two workgroups per CU already overlap each other's staging). fp32 accumulation throughout; Q, K, V, P enter the MFMAs as fp16.
#include <cstdlib>

#include "woq_attn_merge.h"
#include "woq_device.h"
#include "woq_launch.h"

namespace woq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ---- KV element codecs: cache dtype <-> fp16 lanes -------------------------------------------------------------
// eight e4m3 bytes -> eight fp16, exact (every e4m3 value is an fp16): gfx950's packed converter, one instruction per
// pair (round 4; the f32 detour cost three per pair, ~100 VALU per 32-position sub-tile of the long-context decode)
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ _Float16 __attribute__((ext_vector_type(8))) fp8x8_to_h8(
    unsigned int __attribute__((ext_vector_type(2))) raw) {
  const h2v a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw.x, 1.0f, false);
  const h2v b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw.x, 1.0f, true);
  const h2v c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw.y, 1.0f, false);
  const h2v d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw.y, 1.0f, true);
  return (_Float16 __attribute__((ext_vector_type(8)))){a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}
// 8 consecutive cache elements -> 8 fp16
template <int KVD>
__device__ __forceinline__ h8 kv_load8(const void* base, size_t elem) {
  if constexpr (KVD == WOQ_F16) {
    return *(const h8*)((const _Float16*)base + elem);
  } else if constexpr (KVD == WOQ_FP8_E4M3) {
    const u32x2 raw = *(const u32x2*)((const uint8_t*)base + elem);
    return fp8x8_to_h8(raw);
  } else {
    const u32x4 raw = *(const u32x4*)((const uint16_t*)base + elem);
    h8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (_Float16)bf16_bits_to_f32((uint16_t)(raw[j >> 1] >> (16 * (j & 1))));
    return r;
  }
}
// ---- embedding rows: h[m][:] = embed[token[m]][:] (fp32 residual stream) ----------------------------------------
__global__ __launch_bounds__(256) void embed_rows_kernel(const void* __restrict__ embed, int dtype,
                                                         const int32_t* __restrict__ tokens, int hidden,
                                                         float* __restrict__ out) {
  const int m = blockIdx.x;
  const size_t src = (size_t)tokens[m] * hidden;
  for (int i = threadIdx.x; i < hidden; i += 256) out[(size_t)m * hidden + i] = load_f32(embed, src + i, dtype);
}

// ---- RoPE (HF rotate_half form) on q and k, KV append ------------------------------------------------------------
// qkv fp16 [M][(heads + 2 kv_heads) * HD]: q rotated in place; k rotated -> K cache; v -> V cache.
// Row m = sequence m / T, position start + m % T. One thread = 8 consecutive d of the first half of one head slot
// (and their partners in the second half): 16-byte loads and stores throughout.
template <int KVD>
__device__ __forceinline__ void kv_store8(void* base, size_t elem, const float (&v)[8]) {
  if constexpr (KVD == WOQ_FP8_E4M3) {
    u32x2 w8;
    w8.x = f32x2_to_fp8x2(v[0], v[1]) | (f32x2_to_fp8x2(v[2], v[3]) << 16);
    w8.y = f32x2_to_fp8x2(v[4], v[5]) | (f32x2_to_fp8x2(v[6], v[7]) << 16);
    *(u32x2*)((uint8_t*)base + elem) = w8;
    return;
  }
  u32x4 w;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t lo, hi;
    if constexpr (KVD == WOQ_F16) {
      lo = f32_to_f16_bits(fminf(fmaxf(v[2 * j], -65504.f), 65504.f));
      hi = f32_to_f16_bits(fminf(fmaxf(v[2 * j + 1], -65504.f), 65504.f));
    } else {
      lo = f32_to_bf16_bits(v[2 * j]);
      hi = f32_to_bf16_bits(v[2 * j + 1]);
    }
    w[j] = lo | (hi << 16);
  }
  *(u32x4*)((uint16_t*)base + elem) = w;
}

template <int KVD>
__global__ __launch_bounds__(256) void rope_append_kernel(_Float16* __restrict__ qkv, int M, int T, int start,
                                                          int heads, int kv_heads, int HD,
                                                          const float* __restrict__ cs, const float* __restrict__ sn,
                                                          void* __restrict__ kcache, void* __restrict__ vcache,
                                                          size_t seq_stride_elems) {
  const int half = HD >> 1, cph = half >> 3;  // 8-element chunks per half head
  const int nslots = heads + 2 * kv_heads;
  const size_t per_row = (size_t)nslots * cph;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)M * per_row) return;
  const int m = (int)(gid / per_row);
  const int rem = (int)(gid % per_row);
  const int slot = rem / cph, c8 = (rem % cph) * 8;
  const int seq = m / T, pos = start + m % T;
  _Float16* x = qkv + (size_t)m * nslots * HD + (size_t)slot * HD;
  const h8 xa = *(const h8*)(x + c8), xb = *(const h8*)(x + c8 + half);
  float ra[8], rb[8];
  if (slot < heads + kv_heads) {
    const float4_t c0 = *(const float4_t*)(cs + (size_t)pos * half + c8), c1 = *(const float4_t*)(cs + (size_t)pos * half + c8 + 4);
    const float4_t s0 = *(const float4_t*)(sn + (size_t)pos * half + c8), s1 = *(const float4_t*)(sn + (size_t)pos * half + c8 + 4);
    const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = (float)xa[j], b = (float)xb[j];
      ra[j] = a * cc[j] - b * ss[j];
      rb[j] = b * cc[j] + a * ss[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) ra[j] = (float)xa[j], rb[j] = (float)xb[j];
  }
  if (slot < heads) {
    kv_store8<WOQ_F16>(x, c8, ra);
    kv_store8<WOQ_F16>(x, c8 + half, rb);
  } else {
    const int kh = slot < heads + kv_heads ? slot - heads : slot - heads - kv_heads;
    void* cache = slot < heads + kv_heads ? kcache : vcache;
    const size_t e = (size_t)seq * seq_stride_elems + ((size_t)pos * kv_heads + kh) * HD;
    kv_store8<KVD>(cache, e + c8, ra);
    kv_store8<KVD>(cache, e + c8 + half, rb);
  }
}

// ---- causal attention over the cache -----------------------------------------------------------------------------
constexpr int AQB = 128;  // query rows per workgroup
constexpr int AKT = 64;   // cache positions per tile
constexpr int AVRB = 144; // bytes per V^T row: 64 positions x 2 B + 16 pad (conflict-free ds_read_b64, see header)

template <int HD>
__host__ __device__ constexpr int attn_lds_bytes() { return AKT * HD * 2 + HD * AVRB; }

template <int KVD, int HD>
#ifndef WOQ_ATTN_LB
#define WOQ_ATTN_LB 2
#endif
__global__ __launch_bounds__(256, WOQ_ATTN_LB) void attn_prefill_kernel(const _Float16* __restrict__ qkv, int T, int start,
                                                              int heads, int kv_heads, const void* __restrict__ kcache,
                                                              const void* __restrict__ vcache, size_t seq_stride_elems,
                                                              _Float16* __restrict__ out, int n_qblocks, int window,
                                                              int prio) {
  constexpr int CPR = HD / 8;   // 16-B chunks per K row
  constexpr int DC = HD / 32;   // 32-wide d chunks of the QK^T contraction
  constexpr int DT = HD / 16;   // 16-wide d tiles of the output
  constexpr int KRB = HD * 2;   // K tile row bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char asm_raw[];
  unsigned char* ks = asm_raw;                // [64][HD] fp16, chunk slot swizzled
  unsigned char* vs = asm_raw + AKT * KRB;    // [HD][AVRB]: V^T
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  // Which (sequence, head, query block). 3-D grid: x = query block (longest first), y = head, z = sequence. 1-D grid
  // (gridDim.y == 1, heads % 8 == 0): XCD-aware — workgroup ids go round-robin over the 8 XCDs, so id & 7 is the XCD;
  // XCD x takes heads [x heads / 8, (x + 1) heads / 8) of every sequence and walks their query blocks one head after the
  // other. The query blocks of a head (and the query heads of a kv group) then share ONE L2: with the 3-D order a
  // head's 16 blocks ran on all eight XCDs and every L2 had to hold every active head's K / V (4 x 2048 tokens x 32
  // heads = 134 MB per layer, re-read 8.5 times on average: the launch ran at the fabric's ~3.4 TB/s, not on its MFMAs).
  int qb, h, seq;
  if (gridDim.y == 1 && (heads & 7) == 0) {  // a 3-D grid with heads % 8 == 0 has gridDim.y >= 8; heads == 1 is 3-D too
    const int L = (int)blockIdx.x, xcd = L & 7, r = L >> 3;
    const int hp = heads >> 3, g = r / n_qblocks;
    qb = n_qblocks - 1 - r % n_qblocks;
    h = xcd * hp + g % hp;
    seq = g / hp;
  } else {
    qb = n_qblocks - 1 - (int)blockIdx.x;  // longest workgroups first
    h = blockIdx.y, seq = blockIdx.z;
  }
  const int kh = h / (heads / kv_heads);
  const int nslots = heads + 2 * kv_heads;
  const int q0 = qb * AQB + wid * 32;  // this wave's first query row (within the chunk)
  const size_t row_elems = (size_t)nslots * HD;
  const size_t cache0 = (size_t)seq * seq_stride_elems + (size_t)kh * HD;
  const size_t cache_row = (size_t)kv_heads * HD;
  const int kv_len = start + min(qb * AQB + AQB, T);  // positions this workgroup may look at
  const int n_tiles = (kv_len + AKT - 1) / AKT;
  // sliding window (0 = none): query position p sees [p + 1 - window, p]; tiles wholly below the workgroup's first
  // query's window are never loaded
  const int tile0 = window > 0 ? max(0, start + qb * AQB + 1 - window) / AKT : 0;
  const float sc = 1.44269504088896f / sqrtf((float)HD);  // softmax scale in the exp2 domain

  // Q^T fragments: lane (query i16 of row tile rt, quarter kq) holds d = 32c + 8kq .. +8
  h8 qf[2][DC];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int qr = min(q0 + rt * 16 + i16, T - 1);
    const _Float16* qp = qkv + ((size_t)seq * T + qr) * row_elems + (size_t)h * HD;
#pragma unroll
    for (int c = 0; c < DC; ++c) qf[rt][c] = *(const h8*)(qp + c * 32 + kq * 8);
  }
  float4_t o[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) o[dt][rt] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_part[2] = {0.f, 0.f};

  // ---- tile movers. K: thread -> rows tid/CPR + (256/CPR) i, chunk tid % CPR. V: thread -> 4 positions x 8 d ----
  constexpr int KPT = AKT * CPR / 256;  // K vectors per thread
  constexpr int KRS = 256 / CPR;        // K row step between a thread's vectors
  const int k_row = tid / CPR, k_chunk = tid % CPR;
  const int v_g = (tid / CPR), v_c = tid % CPR;  // position quad (0..15 live), d chunk
  const bool v_live = v_g < 16;
  h8 kreg[KPT], vreg[4];
  auto fetch = [&](int tile) {
    const int t0 = tile * AKT;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int t = min(t0 + k_row + i * KRS, kv_len - 1);
      kreg[i] = kv_load8<KVD>(kcache, cache0 + (size_t)t * cache_row + k_chunk * 8);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = min(t0 + min(v_g, 15) * 4 + r, kv_len - 1);
      vreg[r] = kv_load8<KVD>(vcache, cache0 + (size_t)t * cache_row + v_c * 8);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int row = k_row + i * KRS;
      const int f = HD == 128 ? (row & 15) : ((row >> 1) & 7);
      *(h8*)(ks + row * KRB + ((k_chunk ^ f) << 4)) = kreg[i];
    }
    if (v_live) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // d = 8 v_c + i: positions 4 v_g .. +3
        const h4 col = {vreg[0][i], vreg[1][i], vreg[2][i], vreg[3][i]};
        *(h4*)(vs + (v_c * 8 + i) * AVRB + ((v_g ^ (v_c & 15)) << 3)) = col;
      }
    }
  };

  fetch(tile0);
  for (int tile = tile0; tile < n_tiles; ++tile) {
    __syncthreads();  // everyone is done with the previous tile
    stage();
    __syncthreads();
    if (tile + 1 < n_tiles) fetch(tile + 1);  // flies under this tile's MFMAs
    const int t0 = tile * AKT;
    if (t0 > start + q0 + 31) continue;  // wholly above this wave's diagonal (wave-uniform)
    if (window > 0 && t0 + AKT - 1 < start + q0 + 1 - window) continue;  // wholly below this wave's windows

    // ---- S^T = K Q^T ----
    if (prio) __builtin_amdgcn_s_setprio(1);  // MFMA phases outrank the other workgroup's softmax on the same SIMD
    float4_t s[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) s[a][rt] = (float4_t){0.f, 0.f, 0.f, 0.f};
      const int row = a * 16 + i16;
      const int f = HD == 128 ? (row & 15) : ((row >> 1) & 7);
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        const h8 kf = *(const h8*)(ks + row * KRB + (((c * 4 + kq) ^ f) << 4));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) s[a][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[rt][c], s[a][rt], 0, 0, 0);
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
    // ---- causal mask + online softmax (query = lane column i16, positions 16a + 4kq + j) ----
    // some position of the tile may exceed some query of the wave, or fall below some query's window
    const bool diag = t0 + AKT - 1 > start + q0 || (window > 0 && t0 < start + q0 + 32 - window);
    h8 pb[2][2];
    if (diag) {  // wave-uniform: the mask work (compare + select per score) only on tiles that touch the diagonal or a
                 // window edge — most tiles of a long prompt are wholly visible
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int qpos = start + q0 + rt * 16 + i16;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int tp = t0 + a * 16 + kq * 4 + j;
            if (tp > qpos || (window > 0 && tp < qpos + 1 - window)) s[a][rt][j] = -INFINITY;
          }
      }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float mx = -INFINITY;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[a][rt][j]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[rt], mx * sc);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;  // a row with nothing visible yet: keep exp2 finite
      const float alpha = __builtin_amdgcn_exp2f(m_run[rt] - m_use);  // raw v_exp_f32: arguments are <= 0
      const bool moved = m_new != m_run[rt];
      m_run[rt] = m_new;
      float ps = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[a][rt][j], sc, -m_use));
          ps += p;
          pb[a >> 1][rt][(a & 1) * 4 + j] = (_Float16)p;
        }
      l_part[rt] = fmaf(l_part[rt], alpha, ps);
      // once the running maxima have settled (most tiles after the first few) alpha is exactly 1 for every query of
      // the wave: skip the rescale of the output accumulators then (wave-uniform branch)
      if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][rt] *= alpha;
      }
    }
    // ---- O^T += V^T P^T ----
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + i16;
        const int sw = (d >> 3) & 15;
        const unsigned char* vrow = vs + d * AVRB;
        const u32x2 lo = *(const u32x2*)(vrow + (((8 * b + kq) ^ sw) << 3));
        const u32x2 hi = *(const u32x2*)(vrow + (((8 * b + 4 + kq) ^ sw) << 3));
        const h8 vf = __builtin_bit_cast(h8, (u32x4){lo.x, lo.y, hi.x, hi.y});
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) o[dt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[b][rt], o[dt][rt], 0, 0, 0);
      }
    if (prio) __builtin_amdgcn_s_setprio(0);
  }
  // ---- finish: divide by the row sums, store fp16 [M][heads * HD] ----
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    float l = l_part[rt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const int qr = q0 + rt * 16 + i16;
    if (qr < T) {
      _Float16* op = out + ((size_t)seq * T + qr) * ((size_t)heads * HD) + (size_t)h * HD + kq * 4;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const float4_t v = o[dt][rt] * inv;
        *(h4*)(op + dt * 16) = (h4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
      }
    }
  }
}

// ---- long-context decode attention for grouped-query models, on the matrix cores -----------------------------------
// One workgroup per (kv head, context slice): the REP query heads that share the kv head are the 16 columns of
// S^T = K Q^T (REP <= 16 live), so every K / V row of the slice leaves HBM / L2 ONCE for all of them — the
// per-query-head sliced kernel of woq_ops.hip pulls the same rows through the CUs' load paths REP times.
// The four waves are independent streams: wave w owns the 32-position sub-tiles w, w + 4, ... of the slice with its
// own online-softmax state, no workgroup barrier inside the loop, one merge through LDS at the end:
//  * K fragments come STRAIGHT from the cache into MFMA operand registers: the contraction runs over d, and any
//    assignment of d to (MFMA, lane quarter, element) is legal as long as Q uses the same one — lane quarter kq takes
//    d = 32 kq + 8 c + e for MFMA c, i.e. 32 contiguous elements of its row;
//  * q is split into fp16 hi + lo (two MFMAs per fragment): the scores carry fp32-class accuracy, the bar the
//    decode parity tests hold (K / V of an fp8 or fp16 cache are exact in fp16);
//  * V goes through a wave-private transposed LDS tile ([d][32 positions], the 4 x 8 register transpose of
//    attn_prefill_kernel) because positions are the contraction index of O^T = V^T P^T; the probabilities are the
//    S^T accumulators repacked in place (positions {4 kq + j, 16 + 4 kq + j} per lane on both operands);
//  * the next sub-tile's loads are in flight while the current one is in the MFMAs.
// The new token's k / v are applied from LDS by wave 0 of the last slice (rounded to the cache dtype like the rows a
// later step reads back) and appended there. Output: un-normalised partials (o[HD], max, sum) per (head, slice) in the
// natural-exp convention attn_combine_kernel merges.
// One register set of the kernel below: a 32-position sub-tile's K fragments (lane: rows a * 16 + i16, 32 contiguous d)
// and V rows (lane: rows 4 v_g + r, 16 contiguous d). fp16 caches hold them as loaded; an fp8 cache keeps the RAW bytes —
// half the registers, 16-byte requests instead of 8-byte ones — and converts a fragment right where it enters an
// MFMA / the LDS transpose (round 4: 272 -> under 256 registers puts two workgroups on a CU instead of one).
template <int KVD>
struct DecRegs {
  h8 kf[2][4], vr[4][2];
  __device__ __forceinline__ void load_k(int a, const void* kc, size_t row) {
#pragma unroll
    for (int c = 0; c < 4; ++c) kf[a][c] = kv_load8<KVD>(kc, row + c * 8);
  }
  __device__ __forceinline__ void load_v(int r, const void* vc, size_t row) {
    vr[r][0] = kv_load8<KVD>(vc, row);
    vr[r][1] = kv_load8<KVD>(vc, row + 8);
  }
  __device__ __forceinline__ h8 k(int a, int c) const { return kf[a][c]; }
  __device__ __forceinline__ h8 v(int r, int hh) const { return vr[r][hh]; }
};
template <>
struct DecRegs<WOQ_FP8_E4M3> {
  u32x4 kf[2][2], vr[4];
  __device__ __forceinline__ void load_k(int a, const void* kc, size_t row) {
    kf[a][0] = *(const u32x4*)((const uint8_t*)kc + row);
    kf[a][1] = *(const u32x4*)((const uint8_t*)kc + row + 16);
  }
  __device__ __forceinline__ void load_v(int r, const void* vc, size_t row) {
    vr[r] = *(const u32x4*)((const uint8_t*)vc + row);
  }
  __device__ __forceinline__ h8 k(int a, int c) const {
    const u32x4& w = kf[a][c >> 1];
    return fp8x8_to_h8((c & 1) ? (u32x2){w.z, w.w} : (u32x2){w.x, w.y});
  }
  __device__ __forceinline__ h8 v(int r, int hh) const {
    const u32x4& w = vr[r];
    return fp8x8_to_h8(hh ? (u32x2){w.z, w.w} : (u32x2){w.x, w.y});
  }
};

constexpr int DST = 32;   // positions per sub-tile
constexpr int DVRB = 80;  // bytes per V^T row: 32 positions x 2 B + 16 pad
template <int HD, int REP>
__host__ __device__ constexpr int attn_dec_lds_bytes() { return 4 * HD * DVRB + 2 * HD * 4 + REP * HD * 4; }

template <int KVD, int HD, int REP>
__global__ __launch_bounds__(256) void attn_decode_mfma_kernel(const float* __restrict__ qkv, void* __restrict__ kcache,
                                                               void* __restrict__ vcache,
                                                               const int32_t* __restrict__ pos_p,
                                                               const float* __restrict__ cs,
                                                               const float* __restrict__ sn, int hkc, int window,
                                                               float* __restrict__ part, int max_rows, AttnMerge mg) {
  // hkc = heads | kv_heads << 8 | (chunk_fixed / 32) << 16: with `window` the 13th and 14th argument dwords, so that
  // everything in front of the first K / V request is PRELOADED — a kernel's first read of its argument segment costs
  // ~1 us inside a replayed graph (profiles/r06c_xqs_stage_stamps.txt), and heads / kv_heads / window / chunk_fixed
  // used to sit behind the 14 preloaded dwords
  const int heads = hkc & 0xff, kv_heads = (hkc >> 8) & 0xff, chunk_fixed = ((hkc >> 16) & 0xffff) * 32;
  static_assert(HD == 128 && REP <= 16, "one 16-column MFMA tile of query heads, head_dim 128");
  static_assert(16 * (HD + 2) * 4 <= HD * DVRB, "the merge record of a wave reuses its V^T tile");
  constexpr int DC = HD / 32, DT = HD / 16, half = HD / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* vw = dsm_raw + wid * (HD * DVRB);    // this wave's V^T tile, later its merge record
  float* kn = (float*)(dsm_raw + 4 * HD * DVRB);      // [HD] new k (rotated, as the cache holds it), [HD] new v
  float* vn = kn + HD;
  float* qs = vn + HD;                                // [REP][HD] rotated query heads
  const int i16 = lane & 15, kq = lane >> 4;
  const int kh = blockIdx.x, ns = (int)gridDim.y, sp = (int)blockIdx.y;
  // Two slice geometries. ADAPTIVE (chunk_fixed == 0, or a sliding window): the cached span is cut into ns even
  // chunks — every address then depends on the position, which is a device-side word: one dependent round trip before
  // the first K / V byte can be asked for. FIXED (round 4): slice sp owns absolute positions [sp * chunk_fixed,
  // (sp + 1) * chunk_fixed) whatever the position is (the last slice also takes whatever lies beyond), so the two
  // sub-tiles every wave starts with are requested BEFORE the position is read; rows at or beyond it hold zeros or
  // older finite values (the cache is zero-filled at creation) and are masked like ragged tails. Slices that lie
  // wholly beyond the position publish an empty partial.
  const bool fixed = chunk_fixed > 0 && window == 0;
  const size_t cache_row = (size_t)kv_heads * HD;
  const int v_g = lane >> 3, v_c = lane & 7;  // V staging: positions 4 v_g .. + 4, d = 16 v_c .. + 16
  size_t cache0 = (size_t)(fixed ? sp * chunk_fixed : 0) * cache_row + (size_t)kh * HD;
  int last = fixed ? max(0, min(chunk_fixed, max_rows - sp * chunk_fixed) - 1) : 0;
  // two register sets: a wave's first two sub-tiles are requested back to back, then set X is refilled for
  // sub-tile n + 8 as soon as sub-tile n has left it (one exposed load latency per wave, not one per sub-tile)
  DecRegs<KVD> setA, setB;
  auto fetch = [&](DecRegs<KVD>& rs, int sub) {
    const int t0 = sub * DST;
#pragma unroll
    for (int a = 0; a < 2; ++a)
      rs.load_k(a, kcache, cache0 + (size_t)min(t0 + a * 16 + i16, last) * cache_row + kq * 32);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      rs.load_v(r, vcache, cache0 + (size_t)min(t0 + v_g * 4 + r, last) * cache_row + v_c * 16);
  };
  const bool early = fixed && sp * chunk_fixed < max_rows;
  if (early) {
    if (wid * DST <= last) fetch(setA, wid);
    if ((wid + 4) * DST <= last) fetch(setB, wid + 4);
  }
  const int apos = pos_p[0];
  const int w_lo = window > 0 ? max(0, apos + 1 - window) : 0;
  const int span = apos - w_lo;
  int t_lo, npos;
  bool incl_new;
  if (fixed) {
    t_lo = sp * chunk_fixed;
    npos = sp == ns - 1 ? max(apos - t_lo, 0) : max(0, min(apos - t_lo, chunk_fixed));
    incl_new = sp == min(apos / chunk_fixed, ns - 1);
  } else {
    const int chunk = (((span + ns - 1) / ns) + 4 * DST - 1) & ~(4 * DST - 1);
    t_lo = w_lo + min(sp * chunk, span);
    npos = min(apos - t_lo, chunk);  // cached positions of this slice
    incl_new = sp == ns - 1;
    cache0 = (size_t)t_lo * cache_row + (size_t)kh * HD;
  }
  const int n_sub = (npos + DST - 1) / DST;
  const int last_early = last;
  last = max(npos - 1, 0);
  const float sc = 1.44269504088896f / sqrtf((float)HD);

  // prologue, request phase (round 4): the new token's q / k / v and its cos / sin rows are asked for BEFORE the K / V
  // sub-tiles of the adaptive geometry — returns come back in order, so behind 16 KB of cache rows per wave the rotation
  // cannot start until the whole first burst has landed. (Measured neutral: 9.49 vs 9.41 us per layer at 8k,
  // profiles/r04j_*; kept because it is never worse.)
  constexpr int QIT = (REP * HD + 255) / 256;
  static_assert(256 % HD == 0, "a thread keeps its d across the query heads it rotates");
  float q_a[QIT], q_b[QIT];
  const float q_co = cs[(size_t)apos * half + (tid & (half - 1))], q_si = sn[(size_t)apos * half + (tid & (half - 1))];
#pragma unroll
  for (int it = 0; it < QIT; ++it) {
    const int idx = min(tid + it * 256, REP * HD - 1);
    const int d = idx & (HD - 1);
    const float* q = qkv + (size_t)kh * REP * HD + (idx - d);
    q_a[it] = q[d], q_b[it] = q[d ^ half];
  }
  float k_a[8], k_b[8], k_co[8], k_si[8], v_in[8];
  if (tid < HD / 8) {
    const float* k = qkv + (size_t)(heads + kh) * HD;
    const float* v = qkv + (size_t)(heads + kv_heads + kh) * HD;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = tid * 8 + j, i = d & (half - 1);
      k_co[j] = cs[(size_t)apos * half + i], k_si[j] = sn[(size_t)apos * half + i];
      k_a[j] = k[d], k_b[j] = k[d ^ half];
      v_in[j] = v[d];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // what the early requests did not cover: everything in the adaptive geometry; in the fixed one the sub-tiles past
  // the slice's own chunk (short chunks, or the last slice's overflow)
  if (wid < n_sub && !(early && wid * DST <= last_early)) fetch(setA, wid);
  if (wid + 4 < n_sub && !(early && (wid + 4) * DST <= last_early)) fetch(setB, wid + 4);

  // prologue through LDS: threads 0..15 build the new k / v of this kv head (rotated, rounded through the cache dtype
  // like the rows a later step reads back; the slice that holds the new position appends them), everyone rotates the REP
  // query heads
  if (tid < HD / 8) {
    float kk[8], vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = tid * 8 + j;
      kk[j] = d < half ? k_a[j] * k_co[j] - k_b[j] * k_si[j] : k_a[j] * k_co[j] + k_b[j] * k_si[j];
      vv[j] = v_in[j];
    }
    alignas(16) unsigned char tmp[32];
    kv_store8<KVD>(tmp, 0, kk);
    const h8 kr = kv_load8<KVD>(tmp, 0);
    kv_store8<KVD>(tmp, 0, vv);
    const h8 vr8 = kv_load8<KVD>(tmp, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) kn[tid * 8 + j] = (float)kr[j], vn[tid * 8 + j] = (float)vr8[j];
    if (incl_new) {
      const size_t e = (size_t)apos * cache_row + (size_t)kh * HD + tid * 8;
      kv_store8<KVD>(kcache, e, kk);
      kv_store8<KVD>(vcache, e, vv);
    }
  }
#pragma unroll
  for (int it = 0; it < QIT; ++it) {
    const int idx = tid + it * 256;
    if (idx < REP * HD) {
      const int d = idx & (HD - 1);
      qs[idx] = d < half ? q_a[it] * q_co - q_b[it] * q_si : q_a[it] * q_co + q_b[it] * q_si;
    }
  }
  __syncthreads();
  // Q^T fragments, fp16 hi + lo: lane (head i16, quarter kq) holds d = 32 kq + 8 c + e
  h8 qh[DC], ql[DC];
#pragma unroll
  for (int c = 0; c < DC; ++c) {
    float qv[8];
    if (i16 < REP) {
      const float* src = qs + i16 * HD + kq * 32 + c * 8;
      *(float4_t*)&qv[0] = *(const float4_t*)src;
      *(float4_t*)&qv[4] = *(const float4_t*)(src + 4);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 hi = (_Float16)qv[e];
      qh[c][e] = hi;
      ql[c][e] = (_Float16)(qv[e] - (float)hi);
    }
  }
  float4_t o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;

  auto process = [&](DecRegs<KVD>& rs, int sub) {
    // this sub-tile: V^T into LDS, K fragments into the score MFMAs; then the refill of the register set
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const h8 v0 = rs.v(0, hh), v1 = rs.v(1, hh), v2 = rs.v(2, hh), v3 = rs.v(3, hh);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int d = v_c * 16 + hh * 8 + i;
        const h4 col = {v0[i], v1[i], v2[i], v3[i]};
        *(h4*)(vw + d * DVRB + ((v_g ^ v_c) << 3)) = col;
      }
    }
    float4_t s[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      s[a] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        const h8 kf = rs.k(a, c);
        s[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, ql[c], s[a], 0, 0, 0);
        s[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qh[c], s[a], 0, 0, 0);
      }
    }
    const int t0 = sub * DST;
    __builtin_amdgcn_sched_barrier(0);  // the loads below reuse the registers the stores / MFMAs above just released
    if (sub + 8 < n_sub) fetch(rs, sub + 8);
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (t0 + a * 16 + kq * 4 + j >= npos) s[a][j] = -INFINITY;
        mx = fmaxf(mx, s[a][j]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sc);  // finite: position t0 of a processed sub-tile is always live
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
    h8 pb;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[a][j], sc, -m_new));
        ps += p;
        pb[a * 4 + j] = (_Float16)p;
      }
    l_part = fmaf(l_part, alpha, ps);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const unsigned char* vrow = vw + (dt * 16 + i16) * DVRB;
      const u32x2 lo = *(const u32x2*)(vrow + ((kq ^ dt) << 3));
      const u32x2 hi = *(const u32x2*)(vrow + (((4 + kq) ^ dt) << 3));
      const h8 vf = __builtin_bit_cast(h8, (u32x4){lo.x, lo.y, hi.x, hi.y});
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb, o[dt] * alpha, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  };
  for (int sub = wid; sub < n_sub; sub += 8) {
    process(setA, sub);
    if (sub + 4 < n_sub) process(setB, sub + 4);
  }
  if (wid == 0 && incl_new) {  // the new position: score from the fragments, value row from LDS
    float d = 0.f;
    const float* qrow = qs + min(i16, REP - 1) * HD + kq * 32;
#pragma unroll 8
    for (int e = 0; e < 32; ++e) d = fmaf(qrow[e], kn[kq * 32 + e], d);
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    const float m_new = fmaxf(m_run, d * sc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    const float p = __builtin_amdgcn_exp2f(fmaf(d, sc, -m_new));
    m_run = m_new;
    l_part = fmaf(l_part, alpha, kq == 0 ? p : 0.f);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] = fmaf(o[dt][j], alpha, p * vn[dt * 16 + kq * 4 + j]);
  }
  // merge the four waves: record = o[16 heads][HD], max[16], sum[16] (fp32) in the wave's own V^T tile
  float l = l_part;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  {
    float* rec = (float*)vw;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *(float4_t*)(rec + i16 * HD + dt * 16 + kq * 4) = o[dt];
    if (kq == 0) {
      rec[16 * HD + i16] = m_run;
      rec[16 * HD + 16 + i16] = l;
    }
  }
  __syncthreads();
  {
    const int d = tid & (HD - 1);
    for (int h = tid / HD; h < REP; h += 256 / HD) {
      float mw[4], m = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        mw[w] = ((const float*)(dsm_raw + w * (HD * DVRB)))[16 * HD + h];
        m = fmaxf(m, mw[w]);
      }
      float acc = 0.f, den = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float* rec = (const float*)(dsm_raw + w * (HD * DVRB));
        const float wt = mw[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw[w] - m);
        acc = fmaf(rec[h * HD + d], wt, acc);
        den = fmaf(rec[16 * HD + 16 + h], wt, den);
      }
      const float m_nat = m == -INFINITY ? -INFINITY : m * 0.6931471805599453f;  // exp2 domain -> natural
      if (mg.part_g != nullptr) {  // all-to-all merge (round 6): tagged granules, finalised below by the slices themselves
        const AttnA2A a2a{mg.part_g, (mg.seq[0] << 6) | (unsigned int)mg.layer, mg.status};
        attn_a2a_publish(a2a, kh * REP + h, sp, HD, d, acc);
        if (d == 0) {
          attn_a2a_publish(a2a, kh * REP + h, sp, HD, HD, m_nat);
          attn_a2a_publish(a2a, kh * REP + h, sp, HD, HD + 1, den);
        }
        continue;
      }
      // publish (woq_attn_merge.h: agent-scope write-through stores; the merging workgroup may sit on another XCD)
      st_agent(attn_part_o(part, kh * REP + h, sp, HD) + d, acc);
      if (d == 0) {
        float* ml = attn_part_ml(part, heads, kh * REP + h, sp, HD);
        st_agent(ml, m_nat);
        st_agent(ml + 1, den);
      }
    }
  }
  if (mg.part_g != nullptr) {
    // the group's REP * 8 output blocks go round the slices: block B = sp + R * ns is DPP row R's — 16 rows per workgroup
    // and pass, as many passes as REP * 8 / ns needs (REP = 8 in two slices: two)
    constexpr int NB = REP * (HD / 16);
    const AttnA2A a2a{mg.part_g, (mg.seq[0] << 6) | (unsigned int)mg.layer, mg.status};
    for (int R0 = 0; sp + R0 * ns < NB; R0 += 16) {
      if (sp + (R0 + wid * 4) * ns >= NB) continue;  // (wave-uniform: no row of this wave has a block in this pass)
      const int B = sp + (R0 + wid * 4 + (lane >> 4)) * ns;
      const int Bc = min(B, NB - 1);
      attn_a2a_finalize<HD, 32>(a2a, kh * REP + Bc / (HD / 16), Bc % (HD / 16), ns, B < NB, [&](float v, int idx) {
        mg.out[idx] = v;
        if (mg.xo.limbs != nullptr) xq_emit16(v, mg.xo, idx >> 4, idx & 15);
      });
    }
    return;
  }
  // the last slice workgroup of this kv head to get here merges the group's REP heads and emits the attention output
  if (mg.counter != nullptr) attn_slices_merge<HD, (REP < 2 ? REP : 2)>(part, heads, kh * REP, REP, ns, mg.counter + kh, mg, (float*)dsm_raw);
}

// last row of every sequence -> dst fp32 [n_seq][hidden]
__global__ __launch_bounds__(256) void gather_last_kernel(const float* __restrict__ h, int T, int hidden,
                                                          float* __restrict__ dst) {
  const int seq = blockIdx.x;
  const float* src = h + ((size_t)seq * T + (T - 1)) * hidden;
  for (int i = threadIdx.x; i < hidden; i += 256) dst[(size_t)seq * hidden + i] = src[i];
}

// ---- host launchers ---------------------------------------------------------------------------------------------
void launch_embed_rows(const void* embed, int dtype, const int32_t* tokens, int M, int hidden, float* out,
                       hipStream_t st) {
  hipLaunchKernelGGL(embed_rows_kernel, dim3(M), dim3(256), 0, st, embed, dtype, tokens, hidden, out);
}

int launch_rope_append(_Float16* qkv, int n_seq, int T, int start, int heads, int kv_heads, int HD, const float* cs,
                       const float* sn, void* kcache, void* vcache, int kv_dtype, size_t seq_stride_elems,
                       hipStream_t st) {
  if ((HD & 15) != 0) return woq::fail("QBits: head_dim must be a multiple of 16");
  const int M = n_seq * T;
  const size_t threads = (size_t)M * (heads + 2 * kv_heads) * (HD / 16);
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (kv_dtype == WOQ_F16)
    hipLaunchKernelGGL(rope_append_kernel<WOQ_F16>, grid, dim3(256), 0, st, qkv, M, T, start, heads, kv_heads, HD, cs,
                       sn, kcache, vcache, seq_stride_elems);
  else if (kv_dtype == WOQ_BF16)
    hipLaunchKernelGGL(rope_append_kernel<WOQ_BF16>, grid, dim3(256), 0, st, qkv, M, T, start, heads, kv_heads, HD, cs,
                       sn, kcache, vcache, seq_stride_elems);
  else if (kv_dtype == WOQ_FP8_E4M3)
    hipLaunchKernelGGL(rope_append_kernel<WOQ_FP8_E4M3>, grid, dim3(256), 0, st, qkv, M, T, start, heads, kv_heads, HD,
                       cs, sn, kcache, vcache, seq_stride_elems);
  else
    return woq::fail("QBits: unsupported KV cache dtype");
  return 0;
}

template <int KVD, int HD>
static int launch_attn_prefill_t(const _Float16* qkv, int n_seq, int T, int start, int heads, int kv_heads,
                                 const void* kcache, const void* vcache, size_t seq_stride_elems, _Float16* out,
                                 int window, hipStream_t st) {
  auto k = attn_prefill_kernel<KVD, HD>;
  const int nqb = (T + AQB - 1) / AQB;
  static const bool xcd_map = [] {  // WOQ_ATTN_XCD=0: the 3-D order (same-box A/B runs)
    const char* e = getenv("WOQ_ATTN_XCD");
    return !(e && e[0] == '0');
  }();
  const long long total = (long long)nqb * heads * n_seq;
  const dim3 grid = (xcd_map && (heads & 7) == 0 && total < (1ll << 31)) ? dim3((unsigned)total)
                                                                        : dim3((unsigned)nqb, (unsigned)heads, (unsigned)n_seq);
  // s_setprio 1 around the two MFMA phases of a tile: the MFMAs of one workgroup outrank the softmax VALU of the other
  // workgroup on the same SIMD, so the two fall into alternating phases instead of queueing behind each other. Same
  // box, 8 layers at 4 x 2048: 24.42 -> 24.30 ms (attention -5 %, profiles/r03as); WOQ_ATTN_PRIO=0 turns it off.
  static const int prio = [] {
    const char* e = getenv("WOQ_ATTN_PRIO");
    return e ? atoi(e) : 1;
  }();
  hipLaunchKernelGGL(k, grid, dim3(256), attn_lds_bytes<HD>(), st, qkv, T, start, heads, kv_heads, kcache, vcache,
                     seq_stride_elems, out, nqb, window, prio);
  return 0;
}

int launch_attn_prefill(const _Float16* qkv, int n_seq, int T, int start, int heads, int kv_heads, int HD,
                        const void* kcache, const void* vcache, int kv_dtype, size_t seq_stride_elems, _Float16* out,
                        int window, hipStream_t st) {
  if (HD != 64 && HD != 128) return woq::fail("QBits: attention head_dim must be 64 or 128");
#define WOQ_ATTN_CASE(KVD)                                                                                          \
  if (kv_dtype == KVD)                                                                                              \
    return HD == 128 ? launch_attn_prefill_t<KVD, 128>(qkv, n_seq, T, start, heads, kv_heads, kcache, vcache,       \
                                                       seq_stride_elems, out, window, st)                          \
                     : launch_attn_prefill_t<KVD, 64>(qkv, n_seq, T, start, heads, kv_heads, kcache, vcache,        \
                                                      seq_stride_elems, out, window, st);
  WOQ_ATTN_CASE(WOQ_F16)
  WOQ_ATTN_CASE(WOQ_BF16)
  WOQ_ATTN_CASE(WOQ_FP8_E4M3)
#undef WOQ_ATTN_CASE
  return woq::fail("QBits: unsupported KV cache dtype");
}

// long-context decode attention of grouped-query models: true when this kernel took the call (head_dim 128, 2 / 4 / 8
// query heads per kv head), false -> the caller uses the per-query-head sliced kernel
// chunk_fixed: 0 = adaptive slices, else positions per slice (multiple of 32) of the position-independent geometry;
// mg.counter != null: the last slice workgroup of a kv head merges (no combine launch needed)
bool launch_attn_decode_mfma(const float* qkv, void* kcache, void* vcache, int kv_dtype, const int32_t* pos,
                             const float* cs, const float* sn, int heads, int kv_heads, int D, int window, int splits,
                             float* part, int chunk_fixed, int max_ctx, const AttnMerge& mg, hipStream_t st) {
  const int rep = kv_heads > 0 ? heads / kv_heads : 0;
  if (D != 128 || splits <= 1 || splits > ATTN_MAX_SLICES || !(rep == 2 || rep == 4 || rep == 8)) return false;
  if (heads > 255 || kv_heads > 255 || chunk_fixed / 32 > 65535) return false;  // the packed argument dword
  if (chunk_fixed % DST != 0) chunk_fixed = 0;
  const dim3 grid((unsigned)kv_heads, (unsigned)splits);
#define WOQ_DEC_CASE(KVD, R)                                                                                       \
  if (kv_dtype == KVD && rep == R) {                                                                               \
    hipLaunchKernelGGL((attn_decode_mfma_kernel<KVD, 128, R>), grid, dim3(256), (attn_dec_lds_bytes<128, R>()), st, \
                       qkv, kcache, vcache, pos, cs, sn, heads | (kv_heads << 8) | ((chunk_fixed / 32) << 16), window, \
                       part, max_ctx, mg);                                                                         \
    return true;                                                                                                   \
  }
  WOQ_DEC_CASE(WOQ_F16, 2) WOQ_DEC_CASE(WOQ_F16, 4) WOQ_DEC_CASE(WOQ_F16, 8)
  WOQ_DEC_CASE(WOQ_FP8_E4M3, 2) WOQ_DEC_CASE(WOQ_FP8_E4M3, 4) WOQ_DEC_CASE(WOQ_FP8_E4M3, 8)
#undef WOQ_DEC_CASE
  return false;
}

// workgroups of the grouped decode attention kernel the chip holds at once (0 = shape not covered): the all-to-all
// merge needs the whole grid resident
int attn_decode_mfma_slots(int kv_dtype, int rep) {
  static int cache[2][9] = {};  // [fp16 | fp8][rep], 0 = not asked yet, -1 = not covered
  const int ci = kv_dtype == WOQ_FP8_E4M3 ? 1 : 0;
  if ((kv_dtype != WOQ_F16 && kv_dtype != WOQ_FP8_E4M3) || rep < 0 || rep > 8) return 0;
  if (cache[ci][rep] != 0) return std::max(cache[ci][rep], 0);
  cache[ci][rep] = -1;
  int per_cu = 0;
#define WOQ_DEC_OCC(KVD, R)                                                                                          \
  if (kv_dtype == KVD && rep == R) {                                                                                 \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_decode_mfma_kernel<KVD, 128, R>, 256,             \
                                                     attn_dec_lds_bytes<128, R>()) != hipSuccess)                    \
      per_cu = 0;                                                                                                    \
  }
  WOQ_DEC_OCC(WOQ_F16, 2) WOQ_DEC_OCC(WOQ_F16, 4) WOQ_DEC_OCC(WOQ_F16, 8)
  WOQ_DEC_OCC(WOQ_FP8_E4M3, 2) WOQ_DEC_OCC(WOQ_FP8_E4M3, 4) WOQ_DEC_OCC(WOQ_FP8_E4M3, 8)
#undef WOQ_DEC_OCC
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  if (per_cu * cus > 0) cache[ci][rep] = per_cu * cus;
  return per_cu * cus;
}

void launch_gather_last(const float* h, int n_seq, int T, int hidden, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(gather_last_kernel, dim3(n_seq), dim3(256), 0, st, h, T, hidden, dst);
}

}  // namespace woq
