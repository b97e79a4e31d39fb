// woq_gemv_xq.hip — host side of the batch-1 decode GEMV over an XQ activation vector (kernel: woq_gemv_xqs.h) and
// the standalone fp32 -> XQ conversion. Reference path replaced: qbits.cpp:113-140 (woq_linear at M = 1).
// The round-2 kernel (every tile requested up front, offset-binary limbs) lives on as the timing twin of
// tools/xq_probe.hip (tools/xq_r02_twin.h); same-box A/B in profiles/r03*_xq_probe.txt.
#include <algorithm>
#include <cstdlib>

#include "woq_gemv_common.h"
#ifdef WOQ_XQS_STAMPS  // measurement build only (tools/xqs_stamps.py): per-(workgroup, wave) wall-clock stamps of the stages
__device__ unsigned long long* g_xqs_probe = nullptr;
#endif
#include "woq_gemv_xqs.h"
#include "woq_xq.h"

#ifdef WOQ_XQS_STAMPS
extern "C" __attribute__((visibility("default"))) int woq_xqs_set_probe(void* buf_dev) {
  unsigned long long* p = (unsigned long long*)buf_dev;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_xqs_probe), &p, sizeof(p), 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}
#endif

namespace woq {

struct XqLaunch {
  const void* q;
  const void* scales;
  const void* zp;
  XqPtrs xin;
  int tiles_k, K, N, n_groups, tpg_shift, flags;
  float* out;
  const float* bias;
  const float* residual;
  float eps;
  const float* ssq_in;
  int n_ssq;
  XqPtrs xo;
  const float* next_norm_w;
  float* ssq_out;
  const CommDev* tp;  // tensor parallel: push the outputs (partial sums) into the peers' inboxes from the epilogue
  int nw, grid, kt_begin, kt_count;
  LutArgs lut;  // 4-bit table weight types (ndig > 0)
  int ndig;
};

// window depth by tiles per wave (tools/xq_probe.hip grid, profiles/r03c_xq_probe.txt)
// round 6 (profiles/r06ad_gemv_occupancy_and_window.txt): 6 — same-box A/Bs on the round-6 kernel read 4 / 6 / 8 -> 993.5 / 1001.5 / 992.6 tokens/s and
// 4 / 5 / 7 -> 999.8 / 1002.7 / 1003.2 (round 4 had measured no difference); it only reaches the 8-tile waves (o, down)
#ifndef WOQ_XQS_DEPTH  // A/B builds: tools/mkvariant_xq.sh -DWOQ_XQS_DEPTH=4 | 8 (all of a wave's tiles up front)
#define WOQ_XQS_DEPTH 6
#endif
template <int TPW>
struct XqsDepth {
  static constexpr int value = TPW >= WOQ_XQS_DEPTH ? WOQ_XQS_DEPTH : TPW;
};

template <int TPW, int CB, int SMODE, bool ASYM, bool S32, int NDIG>
static int launch_xq_t(const XqLaunch& a, hipStream_t st) {
  typedef XqsLds<TPW, CB, SMODE, ASYM, S32> L;
  const size_t lds = L::total(a.nw);
  if (lds > 160 * 1024) return woq::fail("QBits: XQ GEMV geometry does not fit LDS");
  auto kern = gemv_xqs_kernel<TPW, CB, XqsDepth<TPW>::value, SMODE, ASYM, S32, NDIG>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  const int base = a.kt_count / a.nw, rem = a.kt_count % a.nw;
  XqsLate late;
  late.zp = (const uint8_t*)a.zp, late.xsx = a.xin.sx, late.out = a.out, late.bias = a.bias, late.residual = a.residual;
  late.ssq_in = a.ssq_in, late.next_norm_w = a.next_norm_w, late.ssq_out = a.ssq_out, late.tp = a.tp;
  late.tag_seq = nullptr, late.tag_layer = 0, late.xo = a.xo, late.eps = a.eps, late.N = a.N, late.K = a.K;
  late.n_ssq = a.n_ssq, late.lut = a.lut;
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(a.nw * 64), lds, st, (const u32x4*)a.q, a.scales, a.xin.limbs, a.xin.u,
                     a.tiles_k, a.kt_begin, base, rem, a.n_groups, a.tpg_shift | (a.flags << 8) | (a.nw << 16), late);
  return 0;
}

template <int TPW, int CB>
static int launch_xq_sm(const XqLaunch& a, int smode, bool asym, bool s32, hipStream_t st) {
#define WOQ_XQ_CASE(SM, AS, S3, ND) \
  if (smode == SM && asym == AS && s32 == S3 && a.ndig == ND) return launch_xq_t<TPW, CB, SM, AS, S3, ND>(a, st);
  WOQ_XQ_CASE(0, false, false, 0)
  WOQ_XQ_CASE(0, false, true, 0)
  WOQ_XQ_CASE(0, true, false, 0)
  WOQ_XQ_CASE(0, true, true, 0)
  WOQ_XQ_CASE(1, false, false, 0)
  WOQ_XQ_CASE(1, false, true, 0)
  WOQ_XQ_CASE(1, true, false, 0)
  WOQ_XQ_CASE(1, true, true, 0)
  // 4-bit table types (nf4 / fp4): symmetric; one digit plane (fp4_e2m1), two (bitsandbytes fp4, nf4 at reduced-
  // precision compute) or three (nf4 at compute fp32)
  WOQ_XQ_CASE(0, false, false, 1)
  WOQ_XQ_CASE(0, false, true, 1)
  WOQ_XQ_CASE(1, false, false, 1)
  WOQ_XQ_CASE(1, false, true, 1)
  WOQ_XQ_CASE(0, false, false, 2)
  WOQ_XQ_CASE(0, false, true, 2)
  WOQ_XQ_CASE(1, false, false, 2)
  WOQ_XQ_CASE(1, false, true, 2)
  WOQ_XQ_CASE(0, false, false, 3)
  WOQ_XQ_CASE(0, false, true, 3)
  WOQ_XQ_CASE(1, false, false, 3)
  WOQ_XQ_CASE(1, false, true, 3)
#undef WOQ_XQ_CASE
  return woq::fail("QBits: bad XQ GEMV configuration");
}

// Geometry: nw waves x tpw tiles cover a K range of tiles_k tiles. Measured per projection of the Llama-2-7B layer
// (tools/xq_probe.hip, profiles/r03c_xq_probe.txt): 8 tiles per wave for single column tiles with long K, 4 for the
// fused gate/up pairs (twice the bytes per tile step) and short K. WOQ_XQ_TPW=4|8 forces one (timing experiments).
static bool xq_geometry(int tiles_k, int cb, int smode, int& nw, int& tpw, int ndig = 0) {
  static const int forced = [] {
    const char* s = getenv("WOQ_XQ_TPW");
    return s ? atoi(s) : 0;
  }();
  tpw = (tiles_k > 16 && cb == 1) ? 8 : 4;
  if (forced == 4 || (forced == 8 && !(cb == 2 && smode == 1))) tpw = forced;
  // WOQ_XQ_TPW_SHORT=4|8: single column tiles with K <= 4096 only (o_proj and the stand-alone qkv of Llama-2-7B) —
  // one workgroup per CU there, so more, shorter waves are the other way to hide latency (A/B runs)
  static const int forced_short = [] {
    const char* s = getenv("WOQ_XQ_TPW_SHORT");
    return s ? atoi(s) : 0;
  }();
  if (cb == 1 && tiles_k <= 32 && (forced_short == 4 || forced_short == 8)) tpw = forced_short;
#ifdef WOQ_XQ_TPW6
  static const int forced_long = [] {
    const char* s = getenv("WOQ_XQ_TPW_LONG");
    return s ? atoi(s) : 0;
  }();
  if (cb == 1 && tiles_k > 32 && tiles_k <= 96 && forced_long == 6 && ndig == 0) tpw = 6;
#endif
  const bool wide = cb == 2 && ndig == 3;  // table weights, three digit planes, column-tile pairs: 512-thread launches
  if (wide && tiles_k > 32 && smode == 0) tpw = 8;
  nw = (tiles_k + tpw - 1) / tpw;
  return nw >= 1 && nw <= ((cb * tpw > 8 || wide) ? 8 : 16);  // the kernel's __launch_bounds__
}
// K ranges one launch cannot hold run as chained launches; number of chunks, 0 = not covered
static int xq_k_chunks(int tiles_k, int cb, int smode, bool chainable, int ndig = 0) {
  int nw, tpw;
  if (xq_geometry(tiles_k, cb, smode, nw, tpw, ndig)) return 1;
  if (!chainable) return 0;
  for (int s = 2; s <= 8; ++s)
    if (xq_geometry((tiles_k + s - 1) / s, cb, smode, nw, tpw, ndig)) return s;
  return 0;
}

// does the XQ kernel take this blob as a batch-1 projection? epi 1 = fused gate/up (SiLU * mul)
bool gemv_xq_supported(const woq_blob_header& h, int epi) {
  const bool table = is_table_type(h.weight_type) && h.off_zp == 0;
  if ((h.weight_type != WOQ_W_INT4_CLIP && !table) || h.off_shuffle != 0 || (h.K % WOQ_TILE_K) != 0 || h.K != h.Kpad)
    return false;
  const int tiles_k = h.Kpad / WOQ_TILE_K, cb = epi == 1 ? 2 : 1;
  if (epi == 1 && ((h.Npad / WOQ_TILE_N) & 1)) return false;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return false;
  }
  LutArgs lut;
  return xq_k_chunks(tiles_k, cb, (int)h.scale_mode, epi == 0, lut_args_for(h.weight_type, h.compute_type, lut)) > 0;
}

// out[N] (fp32, may be null when only the XQ output is wanted) = xin . W_deq (* rsqrt(mean(x^2) + eps) when ssq_in)
// (+ bias) (+ residual), SiLU(gate) * up for epi 1; xo (optional): the result as the next kernel's XQ vector, after
// multiplying by next_norm_w (optional), with its per-block sums of squares in ssq_out (optional).
int launch_gemv_xq(const XqPtrs& xin, const void* blob, const woq_blob_header& h, const float* bias, float* out,
                   const float* ssq_in, float eps, const float* residual, int epi, const XqPtrs& xo,
                   const float* next_norm_w, float* ssq_out, hipStream_t st, const CommDev* tp) {
  if (!gemv_xq_supported(h, epi)) return woq::fail("QBits: shape not covered by the XQ GEMV");
  XqLaunch a;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = b + h.off_q;
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.xin = xin;
  a.K = h.K;
  a.N = h.N;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
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
  a.ndig = lut_args_for(h.weight_type, h.compute_type, a.lut);
  a.eps = eps;
  a.n_ssq = h.K / 16;
  if (ssq_in != nullptr && a.n_ssq > 1024) return woq::fail("QBits: RMSNorm partials beyond K = 16384");
  a.next_norm_w = next_norm_w;
  a.ssq_out = ssq_out;
  const int tiles_n = h.Npad / WOQ_TILE_N, cb = epi == 1 ? 2 : 1;
  const int smode = (int)h.scale_mode;
  const bool asym = a.zp != nullptr, s32 = h.scale_type == WOQ_F32;
  const int chunks = xq_k_chunks(a.tiles_k, cb, smode, epi == 0, a.ndig);
  if (chunks > 1 && (ssq_in != nullptr || out == nullptr))
    return woq::fail("QBits: a K range split over chained launches takes no norm and needs an fp32 output");
  a.grid = tiles_n / cb;
  const int per = (a.tiles_k + chunks - 1) / chunks;
  for (int c = 0; c < chunks; ++c) {
    a.kt_begin = c * per;
    a.kt_count = std::min(per, a.tiles_k - a.kt_begin);
    if (a.kt_count <= 0) break;
    const bool last = c == chunks - 1 || a.kt_begin + a.kt_count >= a.tiles_k;
    int tpw;
    if (!xq_geometry(a.kt_count, cb, smode, a.nw, tpw, a.ndig))
      return woq::fail("QBits: shape not covered by the XQ GEMV");
    a.out = out;
    a.bias = c == 0 ? bias : nullptr;
    a.residual = c == 0 ? residual : out;  // chunk c > 0 adds onto the previous chunk's output
    a.ssq_in = ssq_in;
    a.xo = last ? xo : XqPtrs{nullptr, nullptr, nullptr};
    a.tp = last ? tp : nullptr;
    int rc;
    if (cb == 2)
      rc = tpw == 4 ? launch_xq_sm<4, 2>(a, smode, asym, s32, st) : launch_xq_sm<8, 2>(a, smode, asym, s32, st);
    else
#ifdef WOQ_XQ_TPW6  // A/B build (tools/mkvariant_xq.sh t6 -DWOQ_XQ_TPW6=1, WOQ_XQ_TPW_LONG=6): 6-tile waves for long K
      rc = tpw == 6   ? launch_xq_sm<6, 1>(a, smode, asym, s32, st)
           : tpw == 4 ? launch_xq_sm<4, 1>(a, smode, asym, s32, st)
                      : launch_xq_sm<8, 1>(a, smode, asym, s32, st);
#else
      rc = tpw == 4 ? launch_xq_sm<4, 1>(a, smode, asym, s32, st) : launch_xq_sm<8, 1>(a, smode, asym, s32, st);
#endif
    if (rc) return rc;
  }
  return 0;
}

// ---- measurement twins (bench.py roofline.ceiling): what THIS launch structure reaches with the arithmetic taken out --
// load-only twin: the same grid, waves, K slices and non-temporal 16-byte requests as the GEMV of this blob, nothing else
template <int TPW, int CB>
__global__ __launch_bounds__(1024) void gemv_stream_twin_kernel(const u32x4* __restrict__ q, int tiles_k, int kt_off,
                                                                int base_tiles, int rem_tiles,
                                                                unsigned int* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kt0 = kt_off + wid * base_tiles + min(wid, rem_tiles);
  const int cnt = base_tiles + (wid < rem_tiles ? 1 : 0);
  u32x4 acc = {0, 0, 0, 0};
  u32x4 w[CB][TPW];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int kt = min(kt0 + min(t, cnt - 1), tiles_k - 1);
      w[cb][t] = __builtin_nontemporal_load(q + ((size_t)(blockIdx.x * CB + cb) * tiles_k + kt) * 64 + lane);
    }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc |= w[cb][t];
  if ((acc.x | acc.y | acc.z | acc.w) == 0x12345u) sink[0] = 1;  // keeps the loads alive; never true for real blobs
}
__global__ void gemv_empty_twin_kernel(unsigned int* __restrict__ sink) {
  if (threadIdx.x == 0x7fffffffu) sink[0] = 1;
}

// mode 0: load-only twin of the batch-1 GEMV of this blob; mode 1: an empty kernel on the same grid and block
int launch_gemv_twin(const void* blob, const woq_blob_header& h, int epi, int mode, unsigned int* sink, hipStream_t st) {
  const int tiles_k = h.Kpad / WOQ_TILE_K, tiles_n = h.Npad / WOQ_TILE_N, cb = epi == 1 ? 2 : 1;
  LutArgs lut;
  const int ndig = lut_args_for(h.weight_type, h.compute_type, lut);
  const int chunks = xq_k_chunks(tiles_k, cb, (int)h.scale_mode, epi == 0, ndig);
  if (chunks == 0) return woq::fail("QBits: shape not covered by the XQ GEMV");
  const int per = (tiles_k + chunks - 1) / chunks;
  const u32x4* q = (const u32x4*)((const uint8_t*)blob + h.off_q);
  for (int c = 0; c < chunks; ++c) {  // K ranges beyond one launch: the same chained launches as the GEMV
    const int kt_begin = c * per, kt_count = std::min(per, tiles_k - kt_begin);
    if (kt_count <= 0) break;
    int nw, tpw;
    if (!xq_geometry(kt_count, cb, (int)h.scale_mode, nw, tpw, ndig))
      return woq::fail("QBits: shape not covered by the XQ GEMV");
    const int base = kt_count / nw, rem = kt_count % nw;
    const dim3 grid(tiles_n / cb), block(nw * 64);
    if (mode == 1) {
      hipLaunchKernelGGL(gemv_empty_twin_kernel, grid, block, 0, st, sink);
    } else if (cb == 2) {
      if (tpw == 4)
        hipLaunchKernelGGL((gemv_stream_twin_kernel<4, 2>), grid, block, 0, st, q, tiles_k, kt_begin, base, rem, sink);
      else
        hipLaunchKernelGGL((gemv_stream_twin_kernel<8, 2>), grid, block, 0, st, q, tiles_k, kt_begin, base, rem, sink);
    } else {
      if (tpw == 4)
        hipLaunchKernelGGL((gemv_stream_twin_kernel<4, 1>), grid, block, 0, st, q, tiles_k, kt_begin, base, rem, sink);
      else
        hipLaunchKernelGGL((gemv_stream_twin_kernel<8, 1>), grid, block, 0, st, q, tiles_k, kt_begin, base, rem, sink);
    }
  }
  return 0;
}

// ---- standalone conversion: fp32 vector (optionally times a norm weight) -> XQ, one block per 16 threads ----------
__global__ __launch_bounds__(256) void xq_from_f32_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          int K, XqPtrs xo, float* __restrict__ ssq_out) {
  const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;  // K % 16 == 0 and 256 % 16 == 0: rows are whole blocks
  const bool live = i < K;
  const float v = live ? x[i] : 0.f;
  if (ssq_out != nullptr) {
    const float ss = row16_sum(v * v);
    if (live && (i & 15) == 0) ssq_out[i >> 4] = ss;
  }
  const float y = live && g != nullptr ? v * g[i] : v;
  if (i < ((K + 15) & ~15)) xq_emit16(y, xo, i >> 4, i & 15);
}

void launch_xq_from_f32(const float* x, const float* norm_w, int K, const XqPtrs& xo, float* ssq_out, hipStream_t st) {
  hipLaunchKernelGGL(xq_from_f32_kernel, dim3((K + 255) / 256), dim3(256), 0, st, x, norm_w, K, xo, ssq_out);
}

}  // namespace woq
