// woq_gemv.hip — small-M int4 GEMV dispatch + the generic fallback kernel.
//
// Replaces the arithmetic behind qbits.woq_linear at small M:
//   qbits/qbits.cpp:113-140 -> bestla_weightonly_dispatcher.cpp:120-189 (do_compute) ->
//   BesTLA GemmRun<SchedulerBase> with LauncherBase<.., SCoreRowNAvx512f<48,8>, ShuffleActivationKBlockBaseF32,
//   WeightKBlockNInteger, AlphaBetaProcessStoreFp32> [external]: per N-tile x K-block unpack int4 -> fp32,
//   x scale, (asym: - zp * scale * sum(A) via the A-reduce prologue, dispatcher.cpp:154-160), fp32 FMA,
//   epilogue alpha*acc + beta*bias (bestla_customop.hpp:22-40).
// Parity definition: autograd/functions.py:41-63 (dequantize -> [index_select by g_idx] -> fp32 matmul -> + bias).
//
// The per-token hot path is woq_gemv_i8.hip (fp32, aligned, unshuffled activations, K <= 16384: what the decode
// engine and the reference's own fp32 boundary produce). Everything else — bf16 / fp16 activation tensors, the
// GPTQ act-order shuffle (g_idx), unaligned rows, very long K — takes gemv_generic_kernel below: plain fp32 VALU
// arithmetic on the same blob (activation rows staged in LDS as fp32, one workgroup per 16-column tile, waves
// interleaved over the K tiles, wave-shuffle + LDS reduction). It is correct for every blob; it is not tuned.
#include "woq_device.h"
#include "woq_launch.h"

namespace woq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

int gemv_tile_max_rows(const void* act, int act_dtype, int lda, const woq_blob_header& h, const float* norm_w,
                       int epi, int out_dtype);
int launch_gemv_tile(const void* act, int act_dtype, int lda, int M, const void* blob, const woq_blob_header& h,
                     const float* bias, void* out, int out_dtype, int ldo, const float* norm_w, float eps,
                     const float* residual, int ld_res, int epi, hipStream_t st);

struct GenArgs {
  const u32x4* q;
  const u32x4* q_lo;  // fp8 weight types: the LO nibble plane (same tile layout as q = the HI plane), else null
  const void* scales;
  const uint8_t* zp;
  const int32_t* shuffle;
  int K, N, Kpad, tiles_k, n_groups, group, scale_type, scale_mode;
  const void* x;
  int x_dtype, lda, M;
  void* out;
  int out_dtype, ldo, ld_res;
  const float* bias;
  const float* norm_w;
  float eps;
  const float* residual;
  int epi;
  uint32_t weight_type;  // WOQ_W_INT4_CLIP, a 4-bit table type (w = table[code] * scale), or an fp8 type (two planes)
};

constexpr int GEN_NW = 4;    // waves per workgroup
constexpr int GEN_MAXM = 4;  // activation rows per launch

// LDS: [M][Kpad] fp32 activation rows, [GEN_NW][2][GEN_MAXM][16] partial sums, [GEN_NW] reduction scratch
__host__ __device__ inline size_t gen_lds_bytes(int M, int Kpad) {
  return ((size_t)M * Kpad + GEN_NW * 2 * GEN_MAXM * 16 + GEN_NW + 4) * 4;
}

__global__ __launch_bounds__(GEN_NW * 64) void gemv_generic_kernel(GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* xs = gsm;                                // [M][Kpad]
  float* part = xs + (size_t)a.M * a.Kpad;        // [GEN_NW][2][GEN_MAXM][16]
  float* red = part + GEN_NW * 2 * GEN_MAXM * 16;  // [GEN_NW]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int M = a.M, ncb = a.epi == 1 ? 2 : 1;
  __shared__ float lut_s[256];  // table weight types: the 16 values (fp8: all 256), read back by code
  const bool table = is_table_type(a.weight_type);
  const bool fp8 = a.q_lo != nullptr;
  if (fp8)
    lut_s[tid] = fp8_code_value(a.weight_type, tid);  // GEN_NW * 64 = 256 threads
  else if (table && tid < 16)
    lut_s[tid] = lut_value(a.weight_type, tid);

  // ---- stage the activation rows: gather (dtype, shuffle), optional RMSNorm, fp32 in LDS, zero K padding ----
  for (int m = 0; m < M; ++m) {
    const size_t rowoff = (size_t)m * a.lda;
    float ss = 0.f;
    for (int k = tid; k < a.Kpad; k += GEN_NW * 64) {
      float v = 0.f;
      if (k < a.K) v = load_f32(a.x, rowoff + (a.shuffle ? a.shuffle[k] : k), a.x_dtype);
      ss = fmaf(v, v, ss);
      xs[(size_t)m * a.Kpad + k] = v;
    }
    if (a.norm_w) {
      ss = wave_sum(ss);
      if (lane == 0) red[wid] = ss;
      __syncthreads();
      float t = 0.f;
      for (int w2 = 0; w2 < GEN_NW; ++w2) t += red[w2];
      const float inv = 1.0f / sqrtf(t / (float)a.K + a.eps);  // HF LlamaRMSNorm
      for (int k = tid; k < a.K; k += GEN_NW * 64)  // slot k holds x[shuffle[k]] under act-order: its weight goes with it
        xs[(size_t)m * a.Kpad + k] *= inv * a.norm_w[a.shuffle ? a.shuffle[k] : k];
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- inner products: lane (i, kq) owns column i and k = kt*128 + h*64 + kq*16 + j of every tile ----
  for (int cb = 0; cb < ncb; ++cb) {
    const int tn = (int)blockIdx.x * ncb + cb;
    float acc[GEN_MAXM];
#pragma unroll
    for (int m = 0; m < GEN_MAXM; ++m) acc[m] = 0.f;
    for (int kt = wid; kt < a.tiles_k; kt += GEN_NW) {
      const u32x4 wv = a.q[((size_t)tn * a.tiles_k + kt) * 64 + lane];
      u32x4 wl = {0u, 0u, 0u, 0u};
      if (fp8) wl = a.q_lo[((size_t)tn * a.tiles_k + kt) * 64 + lane];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kb = kt * 128 + h * 64 + kq * 16;
        size_t si;
        if (a.scale_mode == 0) {
          int g = kb / a.group;
          if (g >= a.n_groups) g = a.n_groups - 1;
          si = ((size_t)tn * a.n_groups + g) * 16 + i;
        } else {
          si = ((((size_t)tn * a.tiles_k + kt) * 16 + i) << 2) + (2 * h + (kq >> 1));
        }
        const float sc = load_f32(a.scales, si, a.scale_type);
        const int zpv = a.zp ? (int)a.zp[si] - 8 : 0;
        float p[GEN_MAXM];
#pragma unroll
        for (int m = 0; m < GEN_MAXM; ++m) p[m] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const uint32_t word = (j < 8) ? (h == 0 ? wv.x : wv.z) : (h == 0 ? wv.y : wv.w);
          int qv = (int)((word >> nibble_shift(j)) & 0xfu);
          float wq;
          if (fp8) {  // code = (hi nibble << 4) | (lo nibble ^ 8), include/woq_blob.h woq_fp8_headers
            const uint32_t wordl = (j < 8) ? (h == 0 ? wl.x : wl.z) : (h == 0 ? wl.y : wl.w);
            wq = lut_s[(qv << 4) | (int)(((wordl >> nibble_shift(j)) & 0xfu) ^ 8u)];
          } else if (table) {
            wq = lut_s[qv];
          } else {
            qv = (qv & 8) ? qv - 16 : qv;
            wq = (float)(qv - zpv);
          }
#pragma unroll
          for (int m = 0; m < GEN_MAXM; ++m)
            if (m < M) p[m] = fmaf(wq, xs[(size_t)m * a.Kpad + kb + j], p[m]);
        }
#pragma unroll
        for (int m = 0; m < GEN_MAXM; ++m) acc[m] = fmaf(sc, p[m], acc[m]);
      }
    }
#pragma unroll
    for (int m = 0; m < GEN_MAXM; ++m) {
      const float v = reduce_kq(acc[m]);
      if (kq == 0) part[((wid * 2 + cb) * GEN_MAXM + m) * 16 + i] = v;
    }
  }
  __syncthreads();

  // ---- finish: sum over waves, bias, SiLU*mul, residual, store ----
  const int nout = (a.epi == 1 ? 1 : ncb) * M * 16;
  for (int idx = tid; idx < nout; idx += GEN_NW * 64) {
    const int e_i = idx & 15, e_m = (idx >> 4) % M, e_cb = idx / (16 * M);
    float v = 0.f, up = 0.f;
    for (int w2 = 0; w2 < GEN_NW; ++w2) {
      v += part[((w2 * 2 + e_cb) * GEN_MAXM + e_m) * 16 + e_i];
      up += part[((w2 * 2 + 1) * GEN_MAXM + e_m) * 16 + e_i];
    }
    int n;
    if (a.epi == 1) {
      n = (int)blockIdx.x * 16 + e_i;
      if (a.bias) {
        v += a.bias[min(((int)blockIdx.x * 2) * 16 + e_i, a.N - 1)];
        up += a.bias[min(((int)blockIdx.x * 2 + 1) * 16 + e_i, a.N - 1)];
      }
      v = v / (1.0f + __expf(-v)) * up;
    } else {
      n = ((int)blockIdx.x * ncb + e_cb) * 16 + e_i;
      if (a.bias) v += a.bias[min(n, a.N - 1)];
    }
    if (n < (a.epi == 1 ? (a.N >> 1) : a.N)) {
      if (a.residual) v += a.residual[(size_t)e_m * a.ld_res + n];
      store_f32(a.out, (size_t)e_m * a.ldo + n, a.out_dtype, v);
    }
  }
}

static int launch_gemv_generic(const void* act, int act_dtype, int lda, int M, const void* blob,
                               const woq_blob_header& h, const float* bias, void* out, int out_dtype, int ldo,
                               const float* norm_w, float eps, const float* residual, int ld_res, int epi,
                               hipStream_t st, const void* lo_plane = nullptr, uint32_t fp8_type = 0) {
  GenArgs a;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = (const u32x4*)(b + h.off_q);
  a.q_lo = (const u32x4*)lo_plane;
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.shuffle = h.off_shuffle ? (const int32_t*)(b + h.off_shuffle) : nullptr;
  a.K = h.K;
  a.N = h.N;
  a.Kpad = h.Kpad;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
  a.n_groups = h.n_groups;
  a.group = h.group;
  a.scale_type = (int)h.scale_type;
  a.scale_mode = (int)h.scale_mode;
  a.x = act;
  a.x_dtype = act_dtype;
  a.lda = lda;
  a.M = M;
  a.out = out;
  a.out_dtype = out_dtype;
  a.ldo = ldo;
  a.ld_res = ld_res;
  a.bias = bias;
  a.norm_w = norm_w;
  a.eps = eps;
  a.residual = residual;
  a.epi = epi;
  a.weight_type = lo_plane ? fp8_type : h.weight_type;
  const int tiles_n = h.Npad / WOQ_TILE_N;
  if (epi == 1 && (tiles_n & 1)) return woq::fail("QBits: fused gate/up weight needs an even number of column tiles");
  const size_t lds = gen_lds_bytes(M, h.Kpad);
  if (lds > 158 * 1024) return woq::fail("QBits: K too large for the small-M GEMV (activation row does not fit LDS)");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemv_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       158 * 1024);  // the kernel also holds 1 KiB of static LDS (the value table)
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemv_generic_kernel, dim3(tiles_n / (epi == 1 ? 2 : 1)), dim3(GEN_NW * 64), lds, st, a);
  return 0;
}

// Shared by woq_linear and the decode engine: rows in chunks through the i8 tile kernel when the call qualifies,
// through the generic kernel otherwise. `nt` is reserved (weight loads are always non-temporal).
int launch_gemv_from_header(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                            const float* bias, void* out, int out_dtype, int ldo, int M, const float* norm_w,
                            float eps, const float* residual, int ld_res, int epi, int nt, hipStream_t st) {
  (void)nt;
  static const bool force_generic = getenv("WOQ_GEMV_GENERIC") != nullptr;  // A/B switch for tests
  const size_t esz_a = act_dtype == WOQ_F32 ? 4 : 2, esz_o = out_dtype == WOQ_F32 ? 4 : 2;
  int rows = force_generic ? 0 : gemv_tile_max_rows(act, act_dtype, lda, h, norm_w, epi, out_dtype);
  const bool tile = rows > 0 && !(h.off_shuffle != 0 && M > 1);  // act-order blobs: the tile kernel's gather form is batch-1
  if (!tile) {
    rows = GEN_MAXM;
    while (rows > 1 && gen_lds_bytes(rows, h.Kpad) > 150 * 1024) --rows;
  }
  for (int m0 = 0; m0 < M; m0 += rows) {
    const int mc = M - m0 < rows ? M - m0 : rows;
    const void* a_p = (const char*)act + (size_t)m0 * lda * esz_a;
    void* o_p = (char*)out + (size_t)m0 * ldo * esz_o;
    const float* r_p = residual ? residual + (size_t)m0 * ld_res : nullptr;
    // the tile kernel needs every row chunk 16-B aligned, which (lda & 3) == 0 and an aligned base guarantee
    const int rc = tile ? launch_gemv_tile(a_p, act_dtype, lda, mc, blob, h, bias, o_p, out_dtype, ldo, norm_w, eps,
                                           r_p, ld_res, epi, st)
                        : launch_gemv_generic(a_p, act_dtype, lda, mc, blob, h, bias, o_p, out_dtype, ldo, norm_w,
                                              eps, r_p, ld_res, epi, st);
    if (rc) return rc;
  }
  return 0;
}

bool gemv_fp8_mfma_supported(const void* act, int act_dtype, int lda, const woq_blob_header& hi);
int launch_gemv_fp8_mfma(const void* act, int act_dtype, int lda, int M, const void* hi_blob, const woq_blob_header& hi,
                         const void* lo_q, uint32_t fp8_type, const float* bias, void* out, int out_dtype, int ldo,
                         hipStream_t st, const float* norm_w = nullptr, float eps = 0.f, const float* residual = nullptr,
                         int ld_res = 0);

// fp8 weights at decode row counts: the fp8-MFMA kernel (woq_gemv_fp8.hip: code bytes straight into the matrix cores,
// activations as base-16 digits; round 4) where it takes the call, rows in chunks of 8; otherwise (per-32 / per-64 /
// per-256 scales, g_idx, misaligned rows, K beyond 8192) the lookup kernel reading both nibble planes, rows in chunks of GEN_MAXM.
// `hi` = the HI plane's header (scales, shuffle), `lo_q` = the LO plane's qdata.
int launch_gemv_fp8(const void* act, int act_dtype, int lda, const void* hi_blob, const woq_blob_header& hi,
                    const void* lo_q, uint32_t fp8_type, const float* bias, void* out, int out_dtype, int ldo, int M,
                    hipStream_t st) {
  const size_t esz_a = act_dtype == WOQ_F32 ? 4 : 2, esz_o = out_dtype == WOQ_F32 ? 4 : 2;
  if (gemv_fp8_mfma_supported(act, act_dtype, lda, hi)) {
    for (int m0 = 0; m0 < M; m0 += 8) {
      const int rc = launch_gemv_fp8_mfma((const char*)act + (size_t)m0 * lda * esz_a, act_dtype, lda,
                                          M - m0 < 8 ? M - m0 : 8, hi_blob, hi, lo_q, fp8_type, bias,
                                          (char*)out + (size_t)m0 * ldo * esz_o, out_dtype, ldo, st);
      if (rc) return rc;
    }
    return 0;
  }
  int rows = GEN_MAXM;
  while (rows > 1 && gen_lds_bytes(rows, hi.Kpad) > 150 * 1024) --rows;
  for (int m0 = 0; m0 < M; m0 += rows) {
    const int mc = M - m0 < rows ? M - m0 : rows;
    const int rc = launch_gemv_generic((const char*)act + (size_t)m0 * lda * esz_a, act_dtype, lda, mc, hi_blob, hi,
                                       bias, (char*)out + (size_t)m0 * ldo * esz_o, out_dtype, ldo, nullptr, 0.f,
                                       nullptr, 0, 0, st, lo_q, fp8_type);
    if (rc) return rc;
  }
  return 0;
}

void launch_silu_mul_tiles(const float* gu, int inter, float* act, hipStream_t st);

// One batch-1 projection of an fp8-weight layer inside the decode engine (round 6): fp32 activation row in, fp32 out,
// with the engine's fused prologue / epilogues — RMSNorm (norm_w, eps), residual add, and for the fused gate/up
// projection (epi 1) SiLU(gate) * up. The fp8 matrix-core kernel (woq_gemv_fp8.hip) carries the norm and the residual
// itself; its gate/up call writes the 2 * inter interleaved columns to `gu_tmp` and a small launch pairs them. Blobs it
// does not take (g_idx, groups of 256, K beyond 12288) run the lookup kernel, which has all three built in.
int launch_gemv_fp8_engine(const float* act, int lda, const void* hi_blob, const woq_blob_header& hi, const void* lo_q,
                           uint32_t fp8_type, float* out, int ldo, const float* norm_w, float eps, const float* residual,
                           int ld_res, int epi, float* gu_tmp, hipStream_t st) {
  if (gemv_fp8_mfma_supported(act, WOQ_F32, lda, hi) && (epi != 1 || gu_tmp != nullptr)) {
    if (epi == 1) {
      const int rc = launch_gemv_fp8_mfma(act, WOQ_F32, lda, 1, hi_blob, hi, lo_q, fp8_type, nullptr, gu_tmp, WOQ_F32,
                                          hi.N, st, norm_w, eps, nullptr, 0);
      if (rc) return rc;
      launch_silu_mul_tiles(gu_tmp, hi.N / 2, out, st);
      return 0;
    }
    return launch_gemv_fp8_mfma(act, WOQ_F32, lda, 1, hi_blob, hi, lo_q, fp8_type, nullptr, out, WOQ_F32, ldo, st, norm_w,
                                eps, residual, ld_res);
  }
  return launch_gemv_generic(act, WOQ_F32, lda, 1, hi_blob, hi, nullptr, out, WOQ_F32, ldo, norm_w, eps, residual, ld_res,
                             epi, st, lo_q, fp8_type);
}

}  // namespace woq
