/* Plain-C client of the drop-in boundary (include/woq_hip.h, include/woq_blob.h): what a cgo / JNI / FFI binding of
 * the reference's qbits operator would call. Host-only part: sizes a packed weight for every weight type and checks
 * the header arithmetic; with a GPU (argv[1] = "gpu") it also quantises a small weight, runs woq_linear and
 * dequantises. Build: gcc -std=c99 -Iinclude examples/c_client.c -L<dir of libwoq_hip.so> -lwoq_hip [-lamdhip64]
 * The reference side of this call: qbits.cpp:61-140 (pybind11 wrappers over the same operations). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "woq_blob.h"
#include "woq_hip.h"

int main(int argc, char** argv) {
  const int K = 4096, N = 11008, group = 128;
  if (woq_abi_version() != WOQ_ABI_VERSION) return 1;  /* built against another revision of include/woq_hip.h */
  /* host arithmetic shared with the library: header geometry of an int4 sym g128 fp16-scale blob */
  woq_blob_header h;
  if (woq_header_init(&h, K, N, group, WOQ_W_INT4_CLIP, WOQ_F16, WOQ_C_FP32, 0, 0) != 0) return 2;
  size_t sz = woq_packed_weight_size(K, N, group, WOQ_W_INT4_CLIP, WOQ_F16, 0, 0);
  printf("int4 %zu %llu\n", sz, (unsigned long long)h.total_bytes);
  if (sz != h.total_bytes) return 3;
  /* algorithmic bytes: K*N/2 of nibbles + K/group*N fp16 scales, padded sections + 256-B header */
  if (sz < (size_t)K * N / 2 + (size_t)(K / group) * N * 2 + 256) return 4;
  const int types[] = {WOQ_W_INT3_CLIP, WOQ_W_INT2_CLIP, WOQ_W_INT8, WOQ_W_NF4, WOQ_W_FP4_E2M1, WOQ_W_FP4_E2M1_BNB,
                       WOQ_W_FP8_E4M3, WOQ_W_FP8_E5M2};
  for (unsigned i = 0; i < sizeof(types) / sizeof(types[0]); ++i) {
    size_t s = woq_packed_weight_size(K, N, group, types[i], WOQ_F32, 0, 0);
    printf("type %d %zu\n", types[i], s);
    if (s == 0) return 5;
  }
  if (woq_packed_weight_size(K, N, 48, WOQ_W_INT4_CLIP, WOQ_F32, 0, 0) != 0) return 6;          /* bad group */
  if (woq_packed_weight_size(K, N, group, WOQ_W_NF4, WOQ_F32, 1, 0) != 0) return 7;             /* asym table type */
  if (woq_packed_weight_size(K, N, group, WOQ_W_INT4_CLIP, WOQ_SCALE_FP8_E8M0, 0, 0) != 0) return 8;
  if (argc < 2 || strcmp(argv[1], "gpu") != 0) {
    printf("host checks ok\n");
    return 0;
  }
  /* device part. The boundary takes plain device pointers, so the client owns the memory: four HIP runtime entry
   * points, declared here by hand (a C binding of the reference's operator would do the same through its own FFI). */
  printf("devices %d\n", woq_device_count());
  if (woq_device_count() <= 0) return 9;
  {
    extern int hipMalloc(void** p, size_t n);
    extern int hipFree(void* p);
    extern int hipMemcpy(void* dst, const void* src, size_t n, int kind); /* 1 = host->device, 2 = device->host */
    extern int hipDeviceSynchronize(void);
    const int k = 256, n = 64, g = 128, m = 2;
    float* w = (float*)malloc(sizeof(float) * k * n);
    float* x = (float*)malloc(sizeof(float) * m * k);
    float* deq = (float*)malloc(sizeof(float) * k * n);
    float* y = (float*)malloc(sizeof(float) * m * n);
    unsigned seed = 12345u;
    for (int i = 0; i < k * n; ++i) {
      seed = seed * 1664525u + 1013904223u;
      w[i] = ((float)(seed >> 8) / 16777216.0f - 0.5f) * 0.2f;
    }
    for (int i = 0; i < m * k; ++i) {
      seed = seed * 1664525u + 1013904223u;
      x[i] = (float)(seed >> 8) / 16777216.0f - 0.5f;
    }
    woq_blob_header hs;
    if (woq_header_init(&hs, k, n, g, WOQ_W_INT4_CLIP, WOQ_F32, WOQ_C_FP32, 1, 0) != 0) return 10;
    size_t bytes = woq_packed_weight_size(k, n, g, WOQ_W_INT4_CLIP, WOQ_F32, 1, 0);
    void *w_d = NULL, *x_d = NULL, *blob_d = NULL, *deq_d = NULL, *y_d = NULL;
    if (hipMalloc(&w_d, sizeof(float) * k * n) || hipMalloc(&x_d, sizeof(float) * m * k) || hipMalloc(&blob_d, bytes) ||
        hipMalloc(&deq_d, sizeof(float) * k * n) || hipMalloc(&y_d, sizeof(float) * m * n))
      return 11;
    hipMemcpy(w_d, w, sizeof(float) * k * n, 1);
    hipMemcpy(x_d, x, sizeof(float) * m * k, 1);
    /* qbits.quantize_to_packed_weight -> blob (asym int4, group 128, fp32 scales, [K,N] input) */
    if (woq_quantize_to_packed_weight((const float*)w_d, 0, k, n, g, WOQ_W_INT4_CLIP, WOQ_F32, WOQ_C_FP32, 1, blob_d,
                                      bytes, NULL) != 0) {
      printf("quantize: %s\n", woq_last_error());
      return 12;
    }
    woq_blob_header hd;
    if (woq_read_header(blob_d, &hd, NULL) != 0 || hd.K != k || hd.N != n || hd.total_bytes != hs.total_bytes) return 13;
    /* qbits.dequantize_packed_weight and qbits.woq_linear on the same blob */
    if (woq_dequantize_packed_weight(blob_d, &hd, (float*)deq_d, 0, NULL) != 0) return 14;
    if (woq_linear(x_d, WOQ_F32, k, blob_d, &hd, NULL, y_d, WOQ_F32, n, m, NULL) != 0) {
      printf("woq_linear: %s\n", woq_last_error());
      return 15;
    }
    hipDeviceSynchronize();
    hipMemcpy(deq, deq_d, sizeof(float) * k * n, 2);
    hipMemcpy(y, y_d, sizeof(float) * m * n, 2);
    /* the reference's own criterion (qbits_ut/test_weightonly.py:51-88): the op equals activation x dequantised weight;
     * and the dequantised weight is within one quantisation step of the original */
    double worst_w = 0.0, worst_y = 0.0, ymax = 0.0;
    for (int i = 0; i < k * n; ++i) {
      double e = deq[i] - w[i];
      if (e < 0) e = -e;
      if (e > worst_w) worst_w = e;
    }
    for (int r = 0; r < m; ++r)
      for (int c = 0; c < n; ++c) {
        double acc = 0.0;
        for (int i = 0; i < k; ++i) acc += (double)x[r * k + i] * deq[i * n + c];
        double e = acc - y[r * n + c];
        if (e < 0) e = -e;
        if (e > worst_y) worst_y = e;
        if (acc < 0) acc = -acc;
        if (acc > ymax) ymax = acc;
      }
    printf("gpu leg: |deq - w| max %.3e (step %.3e), |woq_linear - x.deq| max %.3e of %.3e\n", worst_w, 0.2 / 15.0,
           worst_y, ymax);
    hipFree(w_d), hipFree(x_d), hipFree(blob_d), hipFree(deq_d), hipFree(y_d);
    free(w), free(x), free(deq), free(y);
    if (worst_w > 0.2 / 15.0 * 0.51 + 1e-6) return 16; /* RTN: half a step of the (max - min) / 15 grid */
    if (worst_y > 1e-4 * ymax + 1e-6) return 17;
  }
  printf("gpu checks ok\n");
  return 0;
}
