// gemm_probe.hip — development probe for the prefill GEMM (not part of the product, not a test): one shape through
// woq_linear of a library given on the command line (so variants built with different -D switches can be compared in
// one GPU visit), timed with HIP events over back-to-back calls, checked on sampled rows against an fp32 product of
// the library's own dequantised weights.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/gemm_probe.hip -o gpurun_out/gemm_probe -ldl
//   gemm_probe <lib.so> [M=8192] [K=4096] [N=22016] [group=128] [asym=0] [compute=bf16|fp32] [reps=20]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "woq_blob.h"
#include "woq_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void fill_uniform(float* p, size_t n, unsigned seed, float lo, float hi) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    p[i] = lo + (hi - lo) * (float)(x >> 8) * (1.f / 16777216.f);
  }
}

// x := fp16-representable values, xh := the same as fp16 (PROBE_ACT16: the activation rows go down as fp16)
__global__ void to_f16(float* x, _Float16* xh, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const _Float16 h = (_Float16)x[i];
    xh[i] = h;
    x[i] = (float)h;
  }
}

// ref[r][n] = sum_k x[rows[r]][k] * w[k][n], fp32, one thread per (r, n)
__global__ void ref_rows(const float* x, const float* w, const int* rows, int K, int N, float* ref) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (n >= N) return;
  const float* xr = x + (size_t)rows[r] * K;
  float s0 = 0.f, s1 = 0.f;
  for (int k = 0; k < K; k += 2) {
    s0 = fmaf(xr[k], w[(size_t)k * N + n], s0);
    s1 = fmaf(xr[k + 1], w[(size_t)(k + 1) * N + n], s1);
  }
  ref[(size_t)r * N + n] = s0 + s1;
}

template <class F>
F sym(void* h, const char* name) {
  void* p = dlsym(h, name);
  if (!p) {
    printf("missing symbol %s\n", name);
    exit(1);
  }
  return (F)p;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    printf("usage: gemm_probe <lib.so> [M K N group asym compute reps]\n");
    return 1;
  }
  const int M = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 4096, N = argc > 4 ? atoi(argv[4]) : 22016;
  const int group = argc > 5 ? atoi(argv[5]) : 128, asym = argc > 6 ? atoi(argv[6]) : 0;
  const bool c32 = argc > 7 && strcmp(argv[7], "fp32") == 0;
  const int reps = argc > 8 ? atoi(argv[8]) : 20;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    printf("dlopen: %s\n", dlerror());
    return 1;
  }
  auto f_size = sym<decltype(&woq_packed_weight_size)>(h, "woq_packed_weight_size");
  auto f_quant = sym<decltype(&woq_quantize_to_packed_weight)>(h, "woq_quantize_to_packed_weight");
  auto f_deq = sym<decltype(&woq_dequantize_packed_weight)>(h, "woq_dequantize_packed_weight");
  auto f_hdr = sym<decltype(&woq_read_header)>(h, "woq_read_header");
  auto f_ws = sym<decltype(&woq_set_workspace)>(h, "woq_set_workspace");
  auto f_lin = sym<decltype(&woq_linear)>(h, "woq_linear");
  auto f_err = sym<decltype(&woq_last_error)>(h, "woq_last_error");
#define WQ(x)                                       \
  do {                                              \
    if ((x) != 0) {                                 \
      printf("woq error: %s (%s)\n", f_err(), #x);  \
      return 1;                                     \
    }                                               \
  } while (0)

  if (getenv("PROBE_DIRTY")) {  // later allocations come out of memory full of NaN patterns
    void* junk;
    const size_t jb = (size_t)64 << 30;
    CK(hipMalloc(&junk, jb));
    CK(hipMemset(junk, 0xFF, jb));
    CK(hipDeviceSynchronize());
    CK(hipFree(junk));
  }
  float *w, *x, *out, *ref;
  CK(hipMalloc(&w, (size_t)K * N * 4));
  CK(hipMalloc(&x, (size_t)M * K * 4));
  CK(hipMalloc(&out, (size_t)M * N * 4));
  fill_uniform<<<2048, 256>>>(w, (size_t)K * N, 11u, asym ? -0.02f : -0.03f, asym ? 0.04f : 0.03f);
  fill_uniform<<<2048, 256>>>(x, (size_t)M * K, 23u, -1.f, 1.f);
  const bool act16 = getenv("PROBE_ACT16") != nullptr;
  _Float16* xh = nullptr;
  if (act16) {
    CK(hipMalloc(&xh, (size_t)M * K * 2));
    to_f16<<<2048, 256>>>(x, xh, (size_t)M * K);
  }
  const void* xin = act16 ? (const void*)xh : (const void*)x;
  const int xdt = act16 ? WOQ_F16 : WOQ_F32;
  const int wt = WOQ_W_INT4_CLIP, sct = WOQ_F16, ct = c32 ? WOQ_C_FP32 : WOQ_C_BF16;
  const size_t bb = f_size(K, N, group, wt, sct, asym, 0);
  if (bb == 0) {
    printf("unsupported geometry\n");
    return 1;
  }
  void* blob;
  CK(hipMalloc(&blob, bb));
  WQ(f_quant(w, 0, K, N, group, wt, sct, ct, asym, blob, bb, nullptr));
  woq_blob_header hdr;
  WQ(f_hdr(blob, &hdr, nullptr));
  WQ(f_deq(blob, &hdr, w, 0, nullptr));  // w := dequantised weights [K][N]
  const size_t ws_bytes = (size_t)(M + 256) * (hdr.Kpad) * 2 * (c32 ? 2 : 1) + (size_t)(M + 256) * 8 + (size_t)hdr.Npad * 4 + (1 << 20);
  void* ws;
  CK(hipMalloc(&ws, ws_bytes));
  WQ(f_ws(ws, ws_bytes));

  std::vector<float> very_first;
  if (getenv("PROBE_FIRST")) {  // the process's very first launch of this GEMM, kept for comparison with a later one
    CK(hipMemset(out, 0xFF, (size_t)M * N * 4));
    WQ(f_lin(xin, xdt, K, blob, &hdr, nullptr, out, WOQ_F32, N, M, nullptr));
    very_first.resize((size_t)M * N);
    CK(hipMemcpy(very_first.data(), out, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  }
  for (int i = 0; i < 3; ++i) WQ(f_lin(xin, xdt, K, blob, &hdr, nullptr, out, WOQ_F32, N, M, nullptr));
  CK(hipDeviceSynchronize());
  if (!very_first.empty()) {
    std::vector<float> now((size_t)M * N);
    CK(hipMemcpy(now.data(), out, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    size_t nd = 0;
    for (size_t i = 0; i < now.size(); ++i)
      if (memcmp(&now[i], &very_first[i], 4) != 0) {
        if (nd < 12)
          printf("    first launch differs: row %zu col %zu (col block %zu wave %zu tile-in-wave %zu i16 %zu): %.6f vs %.6f\n",
                 i / N, i % N, (i % N) / 128, ((i % N) / 32) & 3, ((i % N) / 16) & 1, (i % N) & 15, very_first[i], now[i]);
        ++nd;
      }
    printf("  first launch vs 4th: %zu of %zu elements differ\n", nd, now.size());
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f, total = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) WQ(f_lin(xin, xdt, K, blob, &hdr, nullptr, out, WOQ_F32, N, M, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    best = fminf(best, ms);
    total += ms;
  }
  const double fl = 2.0 * M * (double)K * N;
  printf("%s M=%d K=%d N=%d g=%d asym=%d %s: best %.4f ms (%.1f TFLOP/s, %.3f of 2.5 PF incl. pack pass), mean %.4f ms\n",
         argv[1], M, K, N, group, asym, c32 ? "fp32" : "bf16", best, fl / best / 1e9, fl / best / 1e9 / 2500.0, total / 3);

  if (const char* rp = getenv("PROBE_REPEAT")) {  // determinism: every repeat bit-identical to the first
    const int R = atoi(rp);
    if (getenv("PROBE_OUT16")) {  // fp16 output rows (what the engine's qkv / gate-up calls store)
      std::vector<uint16_t> f16a((size_t)M * N), f16b((size_t)M * N);
      int badr = 0;
      for (int r = 0; r < R; ++r) {
        WQ(f_lin(xin, xdt, K, blob, &hdr, nullptr, out, WOQ_F16, N, M, nullptr));
        CK(hipMemcpy(r == 0 ? f16a.data() : f16b.data(), out, (size_t)M * N * 2, hipMemcpyDeviceToHost));
        if (r && memcmp(f16a.data(), f16b.data(), (size_t)M * N * 2) != 0) ++badr;
      }
      printf("  repeat (fp16 out): %d of %d runs differ from the first\n", badr, R - 1);
    }
    std::vector<float> first((size_t)M * N), cur((size_t)M * N);
    int badruns = 0;
    for (int r = 0; r < R; ++r) {
      if (getenv("PROBE_EVICT")) {  // cold L2 / TLB for the next call
        static float* ev = nullptr;
        if (!ev) CK(hipMalloc(&ev, (size_t)1 << 30));
        fill_uniform<<<4096, 256>>>(ev, (size_t)1 << 28, (unsigned)r, 0.f, 1.f);
      }
      if (getenv("PROBE_CLEAR")) {  // an element the launch does not store stays 0xFFFFFFFF
        CK(hipMemset(out, 0xFF, (size_t)M * N * 4));
        CK(hipDeviceSynchronize());
      }
      WQ(f_lin(xin, xdt, K, blob, &hdr, nullptr, out, WOQ_F32, N, M, nullptr));
      CK(hipMemcpy(r == 0 ? first.data() : cur.data(), out, (size_t)M * N * 4, hipMemcpyDeviceToHost));
      if (getenv("PROBE_CLEAR")) {
        const std::vector<float>& v = r == 0 ? first : cur;
        size_t unw = 0;
        for (size_t i = 0; i < v.size(); ++i) {
          uint32_t b;
          memcpy(&b, &v[i], 4);
          unw += b == 0xFFFFFFFFu;
        }
        if (unw) printf("  run %d: %zu elements were never stored\n", r, unw);
      }
      if (r == 0) continue;
      size_t nd = 0;
      for (size_t i = 0; i < (size_t)M * N; ++i)
        if (memcmp(&first[i], &cur[i], 4) != 0) {
          if (nd < 6 && badruns < 4)
            printf("    run %d: row %zu col %zu (tile %zu wave %zu i16 %zu): %.6f vs %.6f\n", r, i / N, i % N, (i % N) / 128,
                   ((i % N) / 32) & 3, (i % N) & 15, cur[i], first[i]);
          ++nd;
        }
      if (nd) {
        ++badruns;
        if (badruns <= 4) printf("  run %d: %zu elements differ from run 0\n", r, nd);
      }
    }
    printf("  repeat: %d of %d runs differ from the first\n", badruns, R - 1);
  }

  // check on sampled rows
  const int NR = 24;
  std::vector<int> rows(NR);
  for (int i = 0; i < NR; ++i) rows[i] = (int)(((long long)i * 2654435761ll + 17) % M);
  rows[0] = 0, rows[1] = M - 1;
  int* drows;
  CK(hipMalloc(&drows, NR * 4));
  CK(hipMemcpy(drows, rows.data(), NR * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&ref, (size_t)NR * N * 4));
  ref_rows<<<dim3((N + 255) / 256, NR), 256>>>(x, w, drows, K, N, ref);
  std::vector<float> hr((size_t)NR * N), ho(N);
  CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
  double maxref = 0, maxerr = 0, csum = 0;
  for (int r = 0; r < NR; ++r) {
    CK(hipMemcpy(ho.data(), out + (size_t)rows[r] * N, (size_t)N * 4, hipMemcpyDeviceToHost));
    for (int n = 0; n < N; ++n) {
      maxref = fmax(maxref, fabs(hr[(size_t)r * N + n]));
      maxerr = fmax(maxerr, fabs(hr[(size_t)r * N + n] - ho[n]));
      csum += ho[n];
    }
  }
  printf("  check: max |err| %.3e, max |ref| %.3e, ratio %.2e (%s), sampled sum %.6f\n", maxerr, maxref, maxerr / maxref,
         maxerr / maxref < (c32 ? 2e-5 : 3e-3) ? "ok" : "BAD", csum);
  return maxerr / maxref < (c32 ? 2e-5 : 3e-3) ? 0 : 2;
}
