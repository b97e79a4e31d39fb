// woq_pack.hip — load-time kernels: repack / RTN-quantize / dequantize / extract for the WQH1 blob.
//
// Replaces (device-side, gfx950):
//   qbits.repack_quantized_weight   qbits/qbits.cpp:61-77 -> dispatcher/src/bestla_packq_impl.cpp:20-41
//   qbits.quantize_to_packed_weight qbits/qbits.cpp:90-100 -> bestla_weightonly_dispatcher.cpp:66-106
//   qbits.dequantize_packed_weight  qbits/qbits.cpp:102-111 -> bestla_weightonly_dispatcher.cpp:46-64
//   acquire_packed_weight_info (SCALE_TENSOR / ZP_TENSOR)     bestla_packq_impl.cpp:190-198
// All HBM-bound byte shuffles; none of them is on the per-token path.
#include "woq_device.h"
#include "woq_launch.h"

namespace woq {

__global__ void write_header_kernel(woq_blob_header h, woq_blob_header* dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = h;
}

// one thread per packed u32 word: gathers 8 int4 values (8 consecutive k of one column) from int8 [K,N].
// Word #w of lane (kq, i): 64-k half h = w >> 1, part p = w & 1 -> k = kt*128 + h*64 + kq*16 + 8p + jj,
// jj in byte jj & 3, low nibble for jj < 4, high nibble otherwise (include/woq_blob.h).
__global__ void repack_q_kernel(const int8_t* __restrict__ q, uint32_t* __restrict__ qd, int K, int N,
                                int tiles_k, size_t n_words) {
  size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  int s = (int)(w & 3);
  int lane = (int)((w >> 2) & 63);
  size_t tile = w >> 8;
  int kt = (int)(tile % (size_t)tiles_k);
  int tn = (int)(tile / (size_t)tiles_k);
  int i = lane & 15, kq = lane >> 4;
  int n = tn * 16 + i;
  int k0 = kt * 128 + (s >> 1) * 64 + kq * 16 + (s & 1) * 8;
  uint32_t word = 0;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    int k = k0 + jj;
    uint32_t u = 0u;  // padding: q = 0
    if (k < K && n < N) u = (uint32_t)q[(size_t)k * N + n] & 0xfu;  // two's-complement nibble
    word |= u << nibble_shift(jj);
  }
  qd[w] = word;
}

// GPTQ g_idx (group id of every K row, act-order) -> activation shuffle indices: position g*blocksize + c holds the
// c-th row (in increasing row order) whose group is g — the reference's convert_idx (qbits_ut/test_packq.py:22-28),
// which BesTLA applies inside repack (the test hands repack the raw g_idx and reads the converted one back,
// test_packq.py:59-64,98-100). One thread per group scans K: O(G*K) reads, load time only.
__global__ void convert_idx_kernel(const int32_t* __restrict__ g_idx, int K, int blocksize, int n_groups,
                                   int32_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  int c = 0;
  for (int i = 0; i < K; ++i)
    if (g_idx[i] == g) {
      const int p = g * blocksize + c;
      if (p < K) out[p] = i;
      ++c;
    }
}

// one thread per stored scale / zero-point element
__global__ void repack_scale_kernel(const float* __restrict__ scale, const int8_t* __restrict__ zp,
                                    woq_blob_header h, uint8_t* __restrict__ blob, size_t n_scale) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_scale) return;
  int tiles_k = h.Kpad / WOQ_TILE_K;
  int tn, i, kb;  // kb = first k of the 32-row block (mode 1) or of the group (mode 0)
  int g;
  if (h.scale_mode == 0) {
    i = (int)(idx & 15);
    g = (int)((idx >> 4) % (size_t)h.n_groups);
    tn = (int)((idx >> 4) / (size_t)h.n_groups);
    kb = g * h.group;
  } else {
    int s = (int)(idx & 3);
    i = (int)((idx >> 2) & 15);
    int kt = (int)((idx >> 6) % (size_t)tiles_k);
    tn = (int)((idx >> 6) / (size_t)tiles_k);
    kb = kt * 128 + s * 32;
    g = kb / h.group;
    if (g >= h.n_groups) g = h.n_groups - 1;
  }
  int n = tn * 16 + i;
  float sc = 0.f;
  uint8_t uz = 8;
  if (n < h.N && kb < h.K) {
    sc = scale[(size_t)g * h.N + n];
    if (zp) uz = (uint8_t)(zp[(size_t)g * h.N + n] + 8);
  }
  store_f32(blob + h.off_scale, idx, (int)h.scale_type, sc);
  if (h.off_zp) (blob + h.off_zp)[idx] = uz;
}

__device__ __forceinline__ void blob_elem(const uint8_t* blob, const woq_blob_header& h, int k, int n, int& u,
                                          int& uz, float& sc) {
  int tiles_k = h.Kpad / WOQ_TILE_K;
  int tn = n >> 4, i = n & 15, kt = k >> 7, r = k & 127;
  int s = r >> 5, hh = r >> 6, kq = (r & 63) >> 4, j = r & 15;
  size_t word = (((size_t)tn * tiles_k + kt) * 64 + (size_t)(kq * 16 + i)) * 4 + hh * 2 + (j >> 3);
  uint32_t w = ((const uint32_t*)(blob + h.off_q))[word];
  int qv = (int)((w >> nibble_shift(j)) & 0xfu);
  if (is_table_type(h.weight_type)) {  // table types: hand the raw code back, no zero point
    u = qv;
    uz = 0;
    sc = 0.f;
  } else
    u = (qv & 8) ? qv + 8 - 16 : qv + 8;  // unsigned-domain value u = q + 8
  size_t si;
  if (h.scale_mode == 0) {
    int g = k / h.group;
    if (g >= h.n_groups) g = h.n_groups - 1;
    si = ((size_t)tn * h.n_groups + g) * 16 + i;
  } else {
    si = ((((size_t)tn * tiles_k + kt) * 16 + i) << 2) + s;
  }
  sc = load_f32(blob + h.off_scale, si, (int)h.scale_type);
  if (!is_table_type(h.weight_type)) uz = h.off_zp ? (int)(blob + h.off_zp)[si] : 8;
}

// w[k][n] = (u - uz) * scale  == (q - zp) * scale, modules.py:264-295
__global__ void dequant_kernel(const uint8_t* __restrict__ blob, woq_blob_header h, float* __restrict__ out,
                               int transpose, int accumulate) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)h.K * h.N;
  if (idx >= total) return;
  int k, n;
  if (transpose) {
    n = (int)(idx / (size_t)h.K);
    k = (int)(idx % (size_t)h.K);
  } else {
    k = (int)(idx / (size_t)h.N);
    n = (int)(idx % (size_t)h.N);
  }
  int u, uz;
  float sc;
  blob_elem(blob, h, k, n, u, uz, sc);
  const float v = is_table_type(h.weight_type) ? lut_value(h.weight_type, u) * sc : (float)(u - uz) * sc;
  out[idx] = accumulate ? out[idx] + v : v;
}

// ---- int8 composite (include/woq_blob.h woq_int8_headers): split q8 / zp8 / scale into the two int4 operands ----
__global__ void split_int8_kernel(const int8_t* __restrict__ q8, size_t n, int8_t* __restrict__ hi,
                                  int8_t* __restrict__ lo) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int v = q8[idx];
  hi[idx] = (int8_t)(v >> 4);
  lo[idx] = (int8_t)((v & 15) - 8);
}
__global__ void split_int8_params_kernel(const float* __restrict__ sc, const int8_t* __restrict__ zp8, size_t n,
                                         float* __restrict__ sc_hi, int8_t* __restrict__ zp_hi,
                                         int8_t* __restrict__ zp_lo) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  sc_hi[idx] = 16.f * sc[idx];
  const int z = zp8 ? (int)zp8[idx] : 0;
  if (zp_hi) zp_hi[idx] = (int8_t)(z >> 4);
  zp_lo[idx] = (int8_t)((z & 15) - 8);
}
// zp8 [G, N] back from the two blobs: 16 zhi + zlo + 8
__global__ void extract_int8_zp_kernel(const uint8_t* __restrict__ bhi, woq_blob_header hh,
                                       const uint8_t* __restrict__ blo, woq_blob_header hl, int8_t* out) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)hl.n_groups * hl.N;
  if (idx >= total) return;
  int g = (int)(idx / (size_t)hl.N), n = (int)(idx % (size_t)hl.N);
  int u, uzh = 8, uzl;
  float sc;
  if (hh.off_zp) blob_elem(bhi, hh, g * hh.group, n, u, uzh, sc);
  blob_elem(blo, hl, g * hl.group, n, u, uzl, sc);
  out[idx] = (int8_t)(16 * (uzh - 8) + (uzl - 8) + 8);
}

// what = WOQ_ACQ_SCALE_TENSOR -> fp32 [G,N]; WOQ_ACQ_ZP_TENSOR -> int8 [G,N] (signed domain)
__global__ void extract_kernel(const uint8_t* __restrict__ blob, woq_blob_header h, int what, void* out) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)h.n_groups * h.N;
  if (idx >= total) return;
  int g = (int)(idx / (size_t)h.N), n = (int)(idx % (size_t)h.N);
  int u, uz;
  float sc;
  blob_elem(blob, h, g * h.group, n, u, uz, sc);
  if (what == WOQ_ACQ_SCALE_TENSOR)
    ((float*)out)[idx] = sc;
  else
    ((int8_t*)out)[idx] = (int8_t)(uz - 8);
}

// RTN: one thread per (group, column). int4 "clip" range (bits = 4) or int8 (bits = 8). Rounding rule documented in DESIGN.md
// (parity unpinned: BesTLA's quantiser is not in /root/reference). rintf = round-half-even.
__global__ void rtn_kernel(const float* __restrict__ w, int transpose, int K, int N, int group, int n_groups,
                           int asym, int bits, int8_t* __restrict__ q, float* __restrict__ scales,
                           int8_t* __restrict__ zp) {
  // bits = 4: 7 / 15 / 8; 8: 127 / 255 / 128; 3: 3 / 7 / 4; 2: 1 / 3 / 2
  const int qmax = (1 << (bits - 1)) - 1, levels = (1 << bits) - 1, off = 1 << (bits - 1);
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_groups * N) return;
  int g = (int)(idx / (size_t)N), n = (int)(idx % (size_t)N);
  int k0 = g * group, k1 = min(k0 + group, K);
  float mx = -INFINITY, mn = INFINITY, amax = 0.f;
  for (int k = k0; k < k1; ++k) {
    float v = transpose ? w[(size_t)n * K + k] : w[(size_t)k * N + n];
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
    amax = fmaxf(amax, fabsf(v));
  }
  float s;
  int z = 0;
  if (!asym) {
    s = amax / (float)qmax;
    if (s == 0.f) s = 1.f;
  } else {
    s = (mx - mn) / (float)levels;
    if (s == 0.f) s = 1.f;
    z = (int)rintf(-mn / s);
    z = max(0, min(levels, z));
  }
  scales[idx] = s;
  if (asym) zp[idx] = (int8_t)(z - off);
  for (int k = k0; k < k1; ++k) {
    float v = transpose ? w[(size_t)n * K + k] : w[(size_t)k * N + n];
    int qi;
    if (!asym) {
      qi = (int)rintf(v / s);
      qi = max(-off, min(qmax, qi));
    } else {
      qi = (int)rintf(v / s) + z;
      qi = max(0, min(levels, qi)) - off;
    }
    q[(size_t)k * N + n] = (int8_t)qi;
  }
}

// RTN onto a 16-entry table: scale = max|w| / table_max per group, code = nearest table entry of w / scale (lowest
// code on ties). Parity unpinned like the integer rule (BesTLA's quantiser is not in the reference tree).
__global__ void rtn_lut_kernel(const float* __restrict__ w, int transpose, int K, int N, int group, int n_groups,
                               uint32_t weight_type, int8_t* __restrict__ q, float* __restrict__ scales) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_groups * N) return;
  int g = (int)(idx / (size_t)N), n = (int)(idx % (size_t)N);
  int k0 = g * group, k1 = min(k0 + group, K);
  float amax = 0.f;
  for (int k = k0; k < k1; ++k) amax = fmaxf(amax, fabsf(transpose ? w[(size_t)n * K + k] : w[(size_t)k * N + n]));
  float s = amax / lut_max(weight_type);
  if (s == 0.f) s = 1.f;
  scales[idx] = s;
  for (int k = k0; k < k1; ++k) {
    const float v = (transpose ? w[(size_t)n * K + k] : w[(size_t)k * N + n]) / s;
    int best = 0;
    float bd = INFINITY;
    for (int c = 0; c < 16; ++c) {
      const float d = fabsf(v - lut_value(weight_type, c));
      if (d < bd) {
        bd = d;
        best = c;
      }
    }
    q[(size_t)k * N + n] = (int8_t)best;
  }
}

// ---- fp8 weights (include/woq_blob.h woq_fp8_headers) ------------------------------------------------------------
// RTN onto the fp8 grid: scale = max|w| / fp8_max per group (e8m0: the next power of two at or above it, so nothing
// leaves the range), code = the finite code whose value is nearest to w / scale in fp32, lowest code on ties (so +0
// beats -0 and a midpoint takes the smaller code). Parity unpinned like the other rules (BesTLA's quantiser is not
// in the reference tree). Load-time kernel: a plain search over the 256 codes.
__global__ void rtn_fp8_kernel(const float* __restrict__ w, int transpose, int K, int N, int group, int n_groups,
                               uint32_t weight_type, int e8m0, int8_t* __restrict__ q, float* __restrict__ scales) {
  __shared__ float val[256];
  val[threadIdx.x & 255] = fp8_code_value(weight_type, threadIdx.x & 255);
  __syncthreads();
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_groups * N) return;
  int g = (int)(idx / (size_t)N), n = (int)(idx % (size_t)N);
  int k0 = g * group, k1 = min(k0 + group, K);
  float amax = 0.f;
  for (int k = k0; k < k1; ++k) amax = fmaxf(amax, fabsf(transpose ? w[(size_t)n * K + k] : w[(size_t)k * N + n]));
  float s = amax / (weight_type == WOQ_W_FP8_E4M3 ? 448.0f : 57344.0f);
  if (s == 0.f) s = 1.f;
  if (e8m0) {
    int e;
    const float f = frexpf(s, &e);  // s = f * 2^e, f in [0.5, 1)
    s = ldexpf(1.0f, f == 0.5f ? e - 1 : e);
  }
  scales[idx] = s;
  for (int k = k0; k < k1; ++k) {
    const float v = (transpose ? w[(size_t)n * K + k] : w[(size_t)k * N + n]) / s;
    int best = 0;
    float bd = INFINITY;
    for (int c = 0; c < 256; ++c) {
      const float t = val[c];
      if (!(fabsf(t) <= 3.0e38f)) continue;  // NaN / inf codes are never produced
      const float d = fabsf(v - t);
      if (d < bd) {
        bd = d;
        best = c;
      }
    }
    q[(size_t)k * N + n] = (int8_t)best;
  }
}

// w[k][n] = value((hi_nibble << 4) | (lo_nibble ^ 8)) * scale, both planes read together
__global__ void dequant_fp8_kernel(const uint8_t* __restrict__ bhi, woq_blob_header hh, const uint8_t* __restrict__ blo,
                                   woq_blob_header hl, uint32_t weight_type, float* __restrict__ out, int transpose) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)hh.K * hh.N;
  if (idx >= total) return;
  int k, n;
  if (transpose) {
    n = (int)(idx / (size_t)hh.K);
    k = (int)(idx % (size_t)hh.K);
  } else {
    k = (int)(idx / (size_t)hh.N);
    n = (int)(idx % (size_t)hh.N);
  }
  int uh, ul, uz;
  float sc, sc2;
  blob_elem(bhi, hh, k, n, uh, uz, sc);   // u = stored signed nibble + 8
  blob_elem(blo, hl, k, n, ul, uz, sc2);
  out[idx] = fp8_code_value(weight_type, ((uh ^ 8) << 4) | ul) * sc;
}

}  // namespace woq

using namespace woq;

static size_t n_scale_elems(const woq_blob_header& h) {
  size_t tiles_n = (size_t)h.Npad / WOQ_TILE_N, tiles_k = (size_t)h.Kpad / WOQ_TILE_K;
  return h.scale_mode == 0 ? tiles_n * (size_t)h.n_groups * 16u : tiles_n * tiles_k * 64u;
}

extern "C" {

size_t woq_packed_weight_size(int K, int N, int blocksize, int weight_type, int scale_type, int asym,
                              int act_shuffle) {
  woq_blob_header h, hi, lo;
  if (woq_weight_narrow_bits((uint32_t)weight_type)) weight_type = WOQ_W_INT4_CLIP;  // int3 / int2 in int4 storage
  if (woq_weight_is_fp8((uint32_t)weight_type)) {  // symmetric only; e8m0 scales are stored as bf16
    if (asym) return 0;
    const uint32_t st = scale_type == WOQ_SCALE_FP8_E8M0 ? (uint32_t)WOQ_BF16 : (uint32_t)scale_type;
    return woq_fp8_headers(&h, &hi, &lo, K, N, blocksize, (uint32_t)weight_type, st, WOQ_C_FP32, act_shuffle) == 0
               ? h.total_bytes
               : 0;
  }
  if (scale_type == WOQ_SCALE_FP8_E8M0) return 0;  // fp8_e8m0 scales go with fp8 weights only
  if (weight_type == WOQ_W_INT8)
    return woq_int8_headers(&h, &hi, &lo, K, N, blocksize, (uint32_t)scale_type, WOQ_C_FP32, asym, act_shuffle) == 0
               ? h.total_bytes
               : 0;
  if (weight_type != WOQ_W_INT4_CLIP && !is_table_type((uint32_t)weight_type)) return 0;
  if (is_table_type((uint32_t)weight_type) && asym) return 0;  // table types are symmetric (no zero points)
  if (woq_header_init(&h, K, N, blocksize, (uint32_t)weight_type, (uint32_t)scale_type, WOQ_C_FP32, asym,
                      act_shuffle) != 0)
    return 0;
  return h.total_bytes;
}

// one int4 blob at `blob` from signed int4 values / fp32 scales / signed zero points / raw g_idx
static int repack_int4(const int8_t* qweight_dev, const float* scale_dev, const int8_t* zp_dev,
                       const int32_t* g_idx_dev, const woq_blob_header& h, uint8_t* blob, hipStream_t st) {
  // section padding must be deterministic (blobs are compared / checksummed byte-wise)
  WOQ_HIP(hipMemsetAsync(blob + h.off_scale, 0, h.total_bytes - h.off_scale, st));
  hipLaunchKernelGGL(write_header_kernel, dim3(1), dim3(64), 0, st, h, (woq_blob_header*)blob);
  int tiles_k = h.Kpad / WOQ_TILE_K;
  size_t n_words = (size_t)(h.Npad / WOQ_TILE_N) * tiles_k * 256u;
  hipLaunchKernelGGL(repack_q_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, qweight_dev,
                     (uint32_t*)(blob + h.off_q), (int)h.K, (int)h.N, tiles_k, n_words);
  size_t ns = n_scale_elems(h);
  hipLaunchKernelGGL(repack_scale_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, scale_dev, zp_dev,
                     h, blob, ns);
  if (g_idx_dev) {
    WOQ_HIP(hipMemsetAsync(blob + h.off_shuffle, 0, (size_t)h.K * 4u, st));
    hipLaunchKernelGGL(convert_idx_kernel, dim3((unsigned)((h.n_groups + 63) / 64)), dim3(64), 0, st, g_idx_dev,
                       (int)h.K, (int)h.group, (int)h.n_groups, (int32_t*)(blob + h.off_shuffle));
  }
  return 0;
}

int woq_repack_quantized_weight(const int8_t* qweight_dev, const float* scale_dev, const int8_t* zp_dev,
                                const int32_t* g_idx_dev, int K, int N, int blocksize, int weight_type,
                                int scale_type, int compute_type, void* blob_dev, size_t blob_bytes,
                                void* stream) {
  WOQ_TRY
  const int narrow = woq_weight_narrow_bits((uint32_t)weight_type);  // int3_clip / int2_clip: int4 storage + a tag
  if (narrow) weight_type = WOQ_W_INT4_CLIP;
  const bool fp8 = woq_weight_is_fp8((uint32_t)weight_type);
  WOQ_CHECK(weight_type == WOQ_W_INT4_CLIP || weight_type == WOQ_W_INT8 || is_table_type((uint32_t)weight_type) || fp8,
            "QBits: unsupported weight_type in repack (int4_clip | int3_clip | int2_clip | int8 | nf4 | fp4_e2m1 | "
            "fp4_e2m1_bnb | fp8_e4m3 | fp8_e5m2)");
  WOQ_CHECK(!((is_table_type((uint32_t)weight_type) || fp8) && zp_dev != nullptr),
            "QBits: float weight types (nf4 / fp4 / fp8) are symmetric: no zero points");
  const bool e8m0 = scale_type == WOQ_SCALE_FP8_E8M0;
  WOQ_CHECK(!e8m0 || fp8, "QBits: fp8_e8m0 scales are only used with fp8_e4m3 / fp8_e5m2 weights");
  if (e8m0) scale_type = WOQ_BF16;  // every power of two in range is an exact bf16
  WOQ_CHECK(scale_type >= WOQ_F32 && scale_type <= WOQ_F16, "QBits: unsupported scale_type");
  WOQ_CHECK(((uintptr_t)blob_dev & 255u) == 0, "QBits: packed-weight buffer must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  uint8_t* blob = (uint8_t*)blob_dev;
  if (fp8) {  // code bytes -> two nibble planes in the int8 composite container, scales s on both
    woq_blob_header h, hi, lo;
    WOQ_CHECK(woq_fp8_headers(&h, &hi, &lo, K, N, blocksize, (uint32_t)weight_type, (uint32_t)scale_type,
                              (uint32_t)compute_type, g_idx_dev != nullptr) == 0,
              "QBits: unsupported blocksize (must be -1 or a multiple of 32)");
    WOQ_CHECK(blob_bytes >= h.total_bytes, "QBits: packed-weight buffer too small");
    if (e8m0) h.flags |= WOQ_FLAG_SCALE_E8M0;
    const size_t kn = (size_t)K * N, gn = (size_t)h.n_groups * N;
    int8_t* tmp = nullptr;  // q_hi [kn] | q_lo [kn] | zp_lo [gn] (zeros)
    WOQ_HIP(hipMallocAsync((void**)&tmp, 2 * kn + gn, st));
    int8_t *q_hi = tmp, *q_lo = tmp + kn, *z_lo = tmp + 2 * kn;
    hipLaunchKernelGGL(split_int8_kernel, dim3((unsigned)((kn + 255) / 256)), dim3(256), 0, st, qweight_dev, kn, q_hi,
                       q_lo);
    WOQ_HIP(hipMemsetAsync(z_lo, 0, gn, st));
    hipLaunchKernelGGL(write_header_kernel, dim3(1), dim3(64), 0, st, h, (woq_blob_header*)blob);
    int rc = repack_int4(q_hi, scale_dev, nullptr, g_idx_dev, hi, blob + h.off_q, st);
    if (rc == 0) rc = repack_int4(q_lo, scale_dev, z_lo, g_idx_dev, lo, blob + h.off_scale, st);
    hipFreeAsync(tmp, st);
    if (rc) return rc;
    WOQ_HIP(hipGetLastError());
    return 0;
  }
  if (weight_type == WOQ_W_INT8) {
    woq_blob_header h, hi, lo;
    WOQ_CHECK(woq_int8_headers(&h, &hi, &lo, K, N, blocksize, (uint32_t)scale_type, (uint32_t)compute_type,
                               zp_dev != nullptr, g_idx_dev != nullptr) == 0,
              "QBits: unsupported blocksize (must be -1 or a multiple of 32)");
    WOQ_CHECK(blob_bytes >= h.total_bytes, "QBits: packed-weight buffer too small");
    const size_t kn = (size_t)K * N, gn = (size_t)h.n_groups * N;
    int8_t* tmp = nullptr;  // q_hi [kn] | q_lo [kn] | zp_hi [gn] | zp_lo [gn] | sc_hi fp32 [gn]
    WOQ_HIP(hipMallocAsync((void**)&tmp, 2 * kn + 2 * gn + 4 * gn + 16, st));
    int8_t *q_hi = tmp, *q_lo = tmp + kn, *z_hi = tmp + 2 * kn, *z_lo = z_hi + gn;
    float* s_hi = (float*)(((uintptr_t)(z_lo + gn) + 15) & ~(uintptr_t)15);
    hipLaunchKernelGGL(split_int8_kernel, dim3((unsigned)((kn + 255) / 256)), dim3(256), 0, st, qweight_dev, kn, q_hi,
                       q_lo);
    hipLaunchKernelGGL(split_int8_params_kernel, dim3((unsigned)((gn + 255) / 256)), dim3(256), 0, st, scale_dev,
                       zp_dev, gn, s_hi, zp_dev ? z_hi : nullptr, z_lo);
    hipLaunchKernelGGL(write_header_kernel, dim3(1), dim3(64), 0, st, h, (woq_blob_header*)blob);
    int rc = repack_int4(q_hi, s_hi, zp_dev ? z_hi : nullptr, g_idx_dev, hi, blob + h.off_q, st);
    if (rc == 0) rc = repack_int4(q_lo, scale_dev, z_lo, g_idx_dev, lo, blob + h.off_scale, st);
    hipFreeAsync(tmp, st);
    if (rc) return rc;
    WOQ_HIP(hipGetLastError());
    return 0;
  }
  woq_blob_header h;
  WOQ_CHECK(woq_header_init(&h, K, N, blocksize, (uint32_t)weight_type, (uint32_t)scale_type,
                            (uint32_t)compute_type, zp_dev != nullptr, g_idx_dev != nullptr) == 0,
            "QBits: unsupported blocksize (must be -1 or a multiple of 32)");
  WOQ_CHECK(blob_bytes >= h.total_bytes, "QBits: packed-weight buffer too small");
  h.narrow_bits = (uint32_t)narrow;
  int rc = repack_int4(qweight_dev, scale_dev, zp_dev, g_idx_dev, h, blob, st);
  if (rc) return rc;
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

int woq_quantize_to_packed_weight(const float* weight_dev, int transpose, int K, int N, int blocksize,
                                  int weight_type, int scale_type, int compute_type, int asym, void* blob_dev,
                                  size_t blob_bytes, void* stream) {
  WOQ_TRY
  const int narrow = woq_weight_narrow_bits((uint32_t)weight_type);
  const bool fp8 = woq_weight_is_fp8((uint32_t)weight_type);
  WOQ_CHECK(narrow || weight_type == WOQ_W_INT4_CLIP || weight_type == WOQ_W_INT8 ||
                is_table_type((uint32_t)weight_type) || fp8,
            "QBits: unsupported weight_type in quantize (int4_clip | int3_clip | int2_clip | int8 | nf4 | fp4_e2m1 | "
            "fp4_e2m1_bnb | fp8_e4m3 | fp8_e5m2)");
  WOQ_CHECK(!((is_table_type((uint32_t)weight_type) || fp8) && asym),
            "QBits: float weight types (nf4 / fp4 / fp8) are symmetric: asym is not supported");
  WOQ_CHECK(scale_type != WOQ_SCALE_FP8_E8M0 || fp8,
            "QBits: fp8_e8m0 scales are only used with fp8_e4m3 / fp8_e5m2 weights");
  int group = (blocksize <= 0 || blocksize > K) ? K : blocksize;  // blocksize -1 -> K (dispatcher.cpp:296)
  int n_groups = (K + group - 1) / group;
  hipStream_t st = (hipStream_t)stream;
  int8_t* q = nullptr;
  float* sc = nullptr;
  int8_t* zp = nullptr;
  WOQ_HIP(hipMalloc((void**)&q, (size_t)K * N));
  WOQ_HIP(hipMalloc((void**)&sc, (size_t)n_groups * N * sizeof(float)));
  if (asym) WOQ_HIP(hipMalloc((void**)&zp, (size_t)n_groups * N));
  size_t nt = (size_t)n_groups * N;
  if (fp8)
    hipLaunchKernelGGL(rtn_fp8_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, weight_dev, transpose, K,
                       N, group, n_groups, (uint32_t)weight_type, scale_type == WOQ_SCALE_FP8_E8M0 ? 1 : 0, q, sc);
  else if (is_table_type((uint32_t)weight_type))
    hipLaunchKernelGGL(rtn_lut_kernel, dim3((unsigned)((nt + 127) / 128)), dim3(128), 0, st, weight_dev, transpose, K,
                       N, group, n_groups, (uint32_t)weight_type, q, sc);
  else
    hipLaunchKernelGGL(rtn_kernel, dim3((unsigned)((nt + 127) / 128)), dim3(128), 0, st, weight_dev, transpose, K, N,
                       group, n_groups, asym, narrow ? narrow : (weight_type == WOQ_W_INT8 ? 8 : 4), q, sc, zp);
  int rc = woq_repack_quantized_weight(q, sc, zp, nullptr, K, N, blocksize, weight_type, scale_type, compute_type,
                                       blob_dev, blob_bytes, stream);
  hipError_t e = hipStreamSynchronize(st);
  hipFree(q);
  hipFree(sc);
  if (zp) hipFree(zp);
  if (rc != 0) return rc;
  WOQ_HIP(e);
  WOQ_END
}

int woq_dequantize_packed_weight(const void* blob_dev, const woq_blob_header* hdr, float* out_dev, int transpose,
                                 void* stream) {
  WOQ_TRY
  WOQ_CHECK(hdr && hdr->magic == WOQ_BLOB_MAGIC, "QBits: not a WQH1 packed weight");
  size_t total = (size_t)hdr->K * hdr->N;
  if (woq_weight_is_fp8(hdr->weight_type)) {
    woq_blob_header o, hi, lo;
    WOQ_CHECK(woq_fp8_headers(&o, &hi, &lo, hdr->K, hdr->N, hdr->group, hdr->weight_type, hdr->scale_type,
                              hdr->compute_type, hdr->off_shuffle != 0) == 0, "QBits: corrupt fp8 header");
    hipLaunchKernelGGL(dequant_fp8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)blob_dev + hdr->off_q, hi, (const uint8_t*)blob_dev + hdr->off_scale, lo,
                       hdr->weight_type, out_dev, transpose);
    WOQ_HIP(hipGetLastError());
    return 0;
  }
  if (hdr->weight_type == WOQ_W_INT8) {  // (hi - zhi) * 16s + (lo - zlo) * s, two fp32 terms
    woq_blob_header o, hi, lo;
    WOQ_CHECK(woq_int8_headers(&o, &hi, &lo, hdr->K, hdr->N, hdr->group, hdr->scale_type, hdr->compute_type,
                               hdr->off_zp != 0, hdr->off_shuffle != 0) == 0, "QBits: corrupt int8 header");
    hipLaunchKernelGGL(dequant_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)blob_dev + hdr->off_q, hi, out_dev, transpose, 0);
    hipLaunchKernelGGL(dequant_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)blob_dev + hdr->off_scale, lo, out_dev, transpose, 1);
    WOQ_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(dequant_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)blob_dev, *hdr, out_dev, transpose, 0);
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

int woq_read_header(const void* blob_dev, woq_blob_header* hdr_out, void* stream) {
  WOQ_TRY
  WOQ_HIP(hipMemcpyAsync(hdr_out, blob_dev, sizeof(woq_blob_header), hipMemcpyDeviceToHost, (hipStream_t)stream));
  WOQ_HIP(hipStreamSynchronize((hipStream_t)stream));
  WOQ_CHECK(hdr_out->magic == WOQ_BLOB_MAGIC, "QBits: not a WQH1 packed weight");
  WOQ_END
}

int woq_blob_extract(const void* blob_dev, const woq_blob_header* hdr, int what, void* out_dev, void* stream) {
  WOQ_TRY
  WOQ_CHECK(hdr && hdr->magic == WOQ_BLOB_MAGIC, "QBits: not a WQH1 packed weight");
  hipStream_t st = (hipStream_t)stream;
  if (hdr->weight_type == WOQ_W_INT8 || woq_weight_is_fp8(hdr->weight_type)) {  // composite: sections of the LO blob
    woq_blob_header o, hi, lo;
    WOQ_CHECK(woq_int8_headers(&o, &hi, &lo, hdr->K, hdr->N, hdr->group, hdr->scale_type, hdr->compute_type,
                               hdr->off_zp != 0, hdr->off_shuffle != 0) == 0, "QBits: corrupt int8 header");
    const uint8_t* bhi = (const uint8_t*)blob_dev + hdr->off_q;
    const uint8_t* blo = (const uint8_t*)blob_dev + hdr->off_scale;
    size_t total = (size_t)hdr->n_groups * hdr->N;
    if (what == WOQ_ACQ_G_IDX) {
      WOQ_CHECK(hdr->off_shuffle != 0, "QBits: not pack g_idx tensor.");
      WOQ_HIP(hipMemcpyAsync(out_dev, blo + lo.off_shuffle, (size_t)hdr->K * 4u, hipMemcpyDeviceToDevice, st));
    } else if (what == WOQ_ACQ_SCALE_TENSOR) {
      hipLaunchKernelGGL(extract_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, blo, lo, what,
                         out_dev);
    } else if (what == WOQ_ACQ_ZP_TENSOR) {
      WOQ_CHECK(hdr->off_zp != 0, "QBits: not pack zero-point tensor.");
      hipLaunchKernelGGL(extract_int8_zp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bhi, hi, blo,
                         lo, (int8_t*)out_dev);
    } else {
      WOQ_FAIL("QBits: unsupported acquire_type");
    }
    WOQ_HIP(hipGetLastError());
    return 0;
  }
  if (what == WOQ_ACQ_G_IDX) {
    WOQ_CHECK(hdr->off_shuffle != 0, "QBits: not pack g_idx tensor.");
    WOQ_HIP(hipMemcpyAsync(out_dev, (const uint8_t*)blob_dev + hdr->off_shuffle, (size_t)hdr->K * 4u,
                           hipMemcpyDeviceToDevice, st));
  } else if (what == WOQ_ACQ_SCALE_TENSOR || what == WOQ_ACQ_ZP_TENSOR) {
    if (what == WOQ_ACQ_ZP_TENSOR) WOQ_CHECK(hdr->off_zp != 0, "QBits: not pack zero-point tensor.");
    size_t total = (size_t)hdr->n_groups * hdr->N;
    hipLaunchKernelGGL(extract_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const uint8_t*)blob_dev, *hdr, what, out_dev);
    WOQ_HIP(hipGetLastError());
  } else {
    WOQ_FAIL("QBits: unsupported acquire_type");
  }
  WOQ_END
}

}  // extern "C"
