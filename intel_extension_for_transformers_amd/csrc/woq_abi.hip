// woq_abi.hip — C-ABI glue: error string, version, and the woq_linear dispatcher.
//
// woq_linear replaces qbits.woq_linear (qbits/qbits.cpp:113-140) and the string-driven template
// selection under it (bestla_weightonly_dispatcher.cpp:230-382: parse_launcher / parse_store /
// parse_activation / parse_weight / parse_gemm_core). Here the "dispatcher" is a few integer
// compares on the cached header: M <= 16 (int4; 8 for the float weight types) -> decode GEMV, else the MFMA GEMM (three-product fp32-class form for
// compute_dtype fp32, single fp16 product for the reduced-precision compute modes).
#include <mutex>

#include <string.h>

#include "woq_device.h"
#include "woq_gemv_common.h"
#include "woq_launch.h"

namespace woq {
std::string& last_error_ref() {
  static thread_local std::string s;
  return s;
}

namespace {
struct Workspace {
  unsigned char* base = nullptr;
  size_t bytes = 0, used = 0;
  hipStream_t last = nullptr;  // stream of the most recent taker: the slices are released at host return, while the
  bool has_last = false;       // kernels that use them are still queued on that stream
  hipEvent_t ev = nullptr;
  std::mutex mu;
} g_ws;
inline size_t ws_round(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace

// The workspace is handed out by a bump pointer and given back when the host call returns — safe only while every
// user queues on ONE stream. A taker on a different stream first makes its stream wait for what the previous stream
// has queued so far (an event), so two streams (or threads) share the scratch in turn instead of at once. While slices
// of ANOTHER stream are still out (used != 0: a concurrent host thread on its own stream) the bump pointer cannot be
// shared — releases would come back out of order and `last` names one stream only — so such a taker gets a private
// stream-ordered allocation. Takers on the SAME stream may interleave freely: their kernels run in stream order.
void* scratch_take(size_t bytes, hipStream_t st, bool* own) {
  const size_t need = ws_round(bytes);
  std::lock_guard<std::mutex> lock(g_ws.mu);
  if (g_ws.base != nullptr && g_ws.used + need <= g_ws.bytes) {
    if (g_ws.has_last && g_ws.last != st && g_ws.used != 0) goto private_alloc;
    if (g_ws.has_last && g_ws.last != st) {
      bool ordered = false;
      if (g_ws.ev == nullptr && hipEventCreateWithFlags(&g_ws.ev, hipEventDisableTiming) != hipSuccess) g_ws.ev = nullptr;
      if (g_ws.ev != nullptr && hipEventRecord(g_ws.ev, g_ws.last) == hipSuccess &&
          hipStreamWaitEvent(st, g_ws.ev, 0) == hipSuccess)
        ordered = true;
      if (!ordered) {  // (e.g. the previous stream is capturing): fall through to a private allocation
        (void)hipGetLastError();
        goto private_alloc;
      }
    }
    g_ws.last = st;
    g_ws.has_last = true;
    void* p = g_ws.base + g_ws.used;
    g_ws.used += need;
    *own = false;
    return p;
  }
private_alloc:
  void* p = nullptr;
  *own = true;
  if (hipMallocAsync(&p, bytes, st) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void scratch_release(void* p, size_t bytes, bool own, hipStream_t st) {
  if (p == nullptr) return;
  if (own) {
    hipFreeAsync(p, st);
  } else {
    std::lock_guard<std::mutex> lock(g_ws.mu);
    g_ws.used -= ws_round(bytes);
  }
}

struct GemvArgs;
int launch_gemv_from_header(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                            const float* bias, void* out, int out_dtype, int ldo, int M, const float* norm_w,
                            float eps, const float* residual, int ld_res, int epi, int nt, hipStream_t st);
int launch_gemm_f16(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                    const float* bias, void* out, int out_dtype, int ldo, int M, const float* norm_w, float eps,
                    const float* residual, int ld_res, int epi, void* ws, int fp32_class, hipStream_t st, const void* fp8_lo = nullptr, uint32_t fp8_type = 0,
                    size_t ws_bytes = 0);
int launch_gemv_fp8(const void* act, int act_dtype, int lda, const void* hi_blob, const woq_blob_header& hi,
                    const void* lo_q, uint32_t fp8_type, const float* bias, void* out, int out_dtype, int ldo, int M,
                    hipStream_t st);
}  // namespace woq

using namespace woq;

extern "C" {

const char* woq_last_error(void) { return last_error_ref().c_str(); }
int woq_abi_version(void) { return WOQ_ABI_VERSION; }
int woq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// one int4 blob: M <= 16 -> decode GEMV, else the MFMA GEMM (woq_gemm_f16.hip): one fp16 product per operand pair for
// the reduced-precision compute modes, the three-product hi + lo form for compute_dtype fp32.
// `residual` (fp32 [M][ld_res]) is added.
static int linear_int4(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                       const float* bias, void* out, int out_dtype, int ldo, int M, const float* residual, int ld_res,
                       hipStream_t st) {
  static const bool gemm_as_gemv = getenv("WOQ_GEMM_AS_GEMV") != nullptr;  // A/B switches for tests
  static const bool gemv_as_gemm = getenv("WOQ_GEMV_AS_GEMM") != nullptr;
  // 4-bit table weights (nf4 / fp4): M <= 8 the generic fp32 kernel (rows in chunks of 4); above, the MFMA GEMM over a
  // pre-dequantised fragment image of the weight (woq_gemm_f16.hip: deq_frag_kernel + gemm_f16frag_kernel)
  static const bool table_generic = getenv("WOQ_TABLE_GENERIC") != nullptr;  // A/B switch: the round-2 behaviour
  // int4: up to 16 rows stay on the decode GEMV (four MFMA row sets over one pass of the weights, woq_gemv_i8.hip): the
  // MFMA GEMM has one 128-row block and a third of the chip's workgroups there (53-93 us at M = 16 for the Llama-2-7B
  // projections against 14-44 us). WOQ_GEMV_MAX_ROWS=8 gives the round-2 seam back (same-box A/B runs).
  static const int gemv_rows = [] {
    const char* e = getenv("WOQ_GEMV_MAX_ROWS");
    const int v = e ? atoi(e) : 16;
    return v >= 1 && v <= 16 ? v : 16;
  }();
  const int seam = is_table_type(h.weight_type) ? 8 : gemv_rows;
  if ((M > seam || gemv_as_gemm) && !gemm_as_gemv && !(is_table_type(h.weight_type) && (table_generic || M <= 8)))
    return launch_gemm_f16(act, act_dtype, lda, blob, h, bias, out, out_dtype, ldo, M, nullptr, 0.f, residual, ld_res, 0,
                           nullptr, h.compute_type == WOQ_C_FP32 ? 1 : 0, st);
  return launch_gemv_from_header(act, act_dtype, lda, blob, h, bias, out, out_dtype, ldo, M, nullptr, 0.f, residual,
                                 ld_res, 0, 1, st);
}

int woq_table_digit_planes(int weight_type, int compute_type, uint32_t planes[3][4], float* wmul) {
  woq::LutArgs L;
  const int nd = woq::lut_args_for((uint32_t)weight_type, (uint32_t)compute_type, L);
  if (planes != nullptr) memcpy(planes, L.d, sizeof(L.d));
  if (wmul != nullptr) *wmul = L.wmul;
  return nd;
}

int woq_set_workspace(void* workspace_dev, size_t bytes) {
  WOQ_TRY
  WOQ_CHECK(workspace_dev != nullptr || bytes == 0, "QBits: null workspace with a size");
  std::lock_guard<std::mutex> lock(g_ws.mu);
  WOQ_CHECK(g_ws.used == 0, "QBits: workspace changed while a call is using it");
  g_ws.base = (unsigned char*)workspace_dev;
  g_ws.bytes = workspace_dev ? bytes : 0;
  g_ws.has_last = false;
  WOQ_END
}

int woq_linear(const void* act_dev, int act_dtype, int lda, const void* blob_dev, const woq_blob_header* hdr,
               const float* bias_dev, void* out_dev, int out_dtype, int ldo, int M, void* stream) {
  WOQ_TRY
  WOQ_CHECK(hdr && hdr->magic == WOQ_BLOB_MAGIC, "QBits: not a WQH1 packed weight");
  WOQ_CHECK(act_dtype >= WOQ_F32 && act_dtype <= WOQ_F16, "QBits: unsupported qbits data type.");
  WOQ_CHECK(out_dtype >= WOQ_F32 && out_dtype <= WOQ_F16, "QBits: unsupported qbits data type.");
  WOQ_CHECK(lda >= hdr->K && ldo >= hdr->N, "QBits: activation/output leading dimension smaller than K/N");
  if (M <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (woq_weight_is_fp8(hdr->weight_type)) {
    // two nibble planes read together by the generic kernel (woq_blob.h woq_fp8_headers); every M, rows in chunks
    woq_blob_header o, hi, lo;
    WOQ_CHECK(woq_fp8_headers(&o, &hi, &lo, hdr->K, hdr->N, hdr->group, hdr->weight_type, hdr->scale_type,
                              hdr->compute_type, hdr->off_shuffle != 0) == 0, "QBits: corrupt fp8 header");
    const uint8_t* bhi = (const uint8_t*)blob_dev + hdr->off_q;
    const uint8_t* blo = (const uint8_t*)blob_dev + hdr->off_scale;
    static const bool fp8_generic = getenv("WOQ_TABLE_GENERIC") != nullptr;  // A/B switch: the round-2 behaviour
    if (M > 8 && !fp8_generic)  // the MFMA GEMM over a pre-dequantised fragment image (woq_gemm_f16.hip)
      rc = launch_gemm_f16(act_dev, act_dtype, lda, bhi, hi, bias_dev, out_dev, out_dtype, ldo, M, nullptr, 0.f, nullptr, 0,
                           0, nullptr, hdr->compute_type == WOQ_C_FP32 ? 1 : 0, st, blo + lo.off_q, hdr->weight_type);
    else
      rc = launch_gemv_fp8(act_dev, act_dtype, lda, bhi, hi, blo + lo.off_q, hdr->weight_type, bias_dev, out_dev,
                           out_dtype, ldo, M, st);
  } else if (hdr->weight_type == WOQ_W_INT8) {
    // (q8 - zp8) s = (hi - zhi) 16s + (lo - zlo) s: two int4 linears on the same activations (woq_blob.h); the
    // first goes to an fp32 scratch, the second adds it (and the bias) and stores the caller's dtype
    woq_blob_header o, hi, lo;
    WOQ_CHECK(woq_int8_headers(&o, &hi, &lo, hdr->K, hdr->N, hdr->group, hdr->scale_type, hdr->compute_type,
                               hdr->off_zp != 0, hdr->off_shuffle != 0) == 0, "QBits: corrupt int8 header");
    bool own = false;
    const size_t tmp_bytes = (size_t)M * hdr->N * sizeof(float);
    float* tmp = (float*)scratch_take(tmp_bytes, st, &own);
    WOQ_CHECK(tmp != nullptr, "QBits: scratch allocation failed");
    rc = linear_int4(act_dev, act_dtype, lda, (const uint8_t*)blob_dev + hdr->off_q, hi, nullptr, tmp, WOQ_F32,
                     hdr->N, M, nullptr, 0, st);
    if (rc == 0)
      rc = linear_int4(act_dev, act_dtype, lda, (const uint8_t*)blob_dev + hdr->off_scale, lo, bias_dev, out_dev,
                       out_dtype, ldo, M, tmp, hdr->N, st);
    scratch_release(tmp, tmp_bytes, own, st);
  } else {
    rc = linear_int4(act_dev, act_dtype, lda, blob_dev, *hdr, bias_dev, out_dev, out_dtype, ldo, M, nullptr, 0, st);
  }
  if (rc != 0) return rc;
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

}  // extern "C"
