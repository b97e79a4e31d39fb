// woq_gemm.hip — prefill-side int4 weight x fp16/bf16 activation GEMM on MFMA (placeholder: routes to
// the GEMV kernel in row chunks until the MFMA kernel lands).
#include "woq_device.h"
#include "woq_launch.h"
namespace woq {
int launch_gemv_from_header(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                            const float* bias, void* out, int out_dtype, int ldo, int M, const float* norm_w,
                            float eps, const float* residual, int ld_res, int epi, int nt, hipStream_t st);
int launch_gemm_mfma(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                     const float* bias, void* out, int out_dtype, int ldo, int M, hipStream_t st) {
  return launch_gemv_from_header(act, act_dtype, lda, blob, h, bias, out, out_dtype, ldo, M, nullptr, 0.f, nullptr, 0,
                                 0, 0, st);
}
}  // namespace woq
