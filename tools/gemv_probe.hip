// gemv_probe.hip — development probe for the decode GEMV (not part of the product, not a test):
// per-shape back-to-back launch time over many distinct blobs (no MALL reuse), a load-only kernel with the
// same decomposition (what the access pattern alone can reach), and a per-stage wall-clock timeline.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DWOQ_PROBE -I include tools/gemv_probe.hip -o gpurun_out/gemv_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

__device__ unsigned long long* g_probe = nullptr;
int g_probe_flags = 0;  // OR-ed into the kernel flags (experiment switches)
#include "../intel_extension_for_transformers_amd/csrc/woq_gemv_i8.hip"

namespace woq {
std::string& last_error_ref() {
  static std::string s;
  return s;
}
}  // namespace woq

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// same decomposition, loads only
template <int TPW, int CB>
__global__ __launch_bounds__(1024) void stream_only(const u32x4* q, int tiles_k, unsigned* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kt0 = (wid * tiles_k) / (int)(blockDim.x >> 6);
  u32x4 acc = {0, 0, 0, 0};
  u32x4 w[CB][TPW];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int kt = min(kt0 + t, tiles_k - 1);
      w[cb][t] = __builtin_nontemporal_load(q + ((size_t)(blockIdx.x * CB + cb) * tiles_k + kt) * 64 + lane);
    }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc |= w[cb][t];
  if ((acc.x | acc.y | acc.z | acc.w) == 0x12345u) sink[0] = 1;
}

__global__ void fill_random(unsigned* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    p[i] = x;
  }
}
__global__ void fill_f32(float* p, size_t n, float scale, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    p[i] = ((float)(x & 0xffff) / 32768.f - 1.f) * scale;
  }
}
__global__ void fill_u16(unsigned short* p, size_t n, unsigned short v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

struct Shape {
  const char* name;
  int K, N, epi, norm, res;
};

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  Shape shapes[] = {{"qkv", 4096, 12288, 0, 1, 0}, {"o", 4096, 4096, 0, 0, 1}, {"gate_up", 4096, 22016, 1, 1, 0},
                    {"down", 11008, 4096, 0, 0, 1}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *x, *g, *res, *out;
  CK(hipMalloc(&x, 16384 * 4));
  CK(hipMalloc(&g, 16384 * 4));
  CK(hipMalloc(&res, 32768 * 4));
  CK(hipMalloc(&out, 32768 * 4));
  fill_f32<<<64, 256, 0, st>>>(x, 16384, 1.f, 1);
  fill_f32<<<64, 256, 0, st>>>(g, 16384, 1.f, 2);
  fill_f32<<<64, 256, 0, st>>>(res, 32768, 1.f, 3);
  unsigned long long* probe;
  const size_t probe_n = (size_t)2048 * 16 * 32;
  CK(hipMalloc(&probe, probe_n * 8));
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  for (const Shape& s : shapes) {
    woq_blob_header h;
    woq_header_init(&h, s.K, s.N, 128, WOQ_W_INT4_CLIP, WOQ_F16, WOQ_C_FP32, 0, 0);
    const size_t bytes = h.total_bytes;
    const int nb = (int)std::max<size_t>(4, (size_t)600e6 / bytes);
    std::vector<unsigned char*> blobs(nb);
    for (int b = 0; b < nb; ++b) {
      CK(hipMalloc(&blobs[b], bytes));
      fill_random<<<1024, 256, 0, st>>>((unsigned*)(blobs[b] + h.off_q), (h.off_scale - h.off_q) / 4, 77u + b);
      fill_u16<<<256, 256, 0, st>>>((unsigned short*)(blobs[b] + h.off_scale), (bytes - h.off_scale) / 2, 0x2000);
    }
    CK(hipStreamSynchronize(st));
    const double alg = (double)s.K * s.N * 0.5 + (double)h.n_groups * s.N * 2;
    auto launch = [&](int b) {
      int rc = woq::launch_gemv_tile(x, WOQ_F32, s.K, 1, blobs[b], h, nullptr, out, WOQ_F32, s.epi ? s.N / 2 : s.N,
                                     s.norm ? g : nullptr, 1e-5f, s.res ? res : nullptr, s.N, s.epi, st);
      if (rc) {
        printf("launch failed: %s\n", woq::last_error_ref().c_str());
        exit(1);
      }
    };
    for (int b = 0; b < nb; ++b) launch(b);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
      for (int b = 0; b < nb; ++b) launch(b);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * nb);
    printf("%-8s K=%5d N=%5d  %6.2f MB  back-to-back %6.2f us/launch  %6.0f GB/s algorithmic\n", s.name, s.K, s.N,
           alg / 1e6, us, alg / us / 1e3);

    // same launch on ONE blob over and over: weights resident in L2 / Infinity Cache (what a perfect warm-up buys)
    {
      for (int r = 0; r < 20; ++r) launch(0);
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < 200; ++r) launch(0);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("         cache-resident weights back-to-back %6.2f us/launch\n", ms * 1e3 / 200);
    }
    // experiment switches (probe build only): what does each stage cost end to end?
    {
      const int exps[] = {16, 32, 64, 128, 256, 32 | 128, 32 | 64 | 128, 32 | 64 | 128 | 256};
      const char* en[] = {"no residual load", "no staging math", "no mfma loop", "no x loads", "no barrier/epilogue",
                          "no staging, no x", "no staging/x/mfma", "loads only"};
      for (int e = 0; e < 8; ++e) {
        g_probe_flags = exps[e];
        for (int b = 0; b < nb; ++b) launch(b);
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r)
          for (int b = 0; b < nb; ++b) launch(b);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("         %-22s back-to-back %6.2f us/launch\n", en[e], ms * 1e3 / (reps * nb));
      }
      g_probe_flags = 0;
    }

    // load-only twin
    const int tiles_k = h.Kpad / 128, tiles_n = h.Npad / 16;
    const int cb = s.epi ? 2 : 1;
    auto launch_so = [&](int b) {
      const u32x4* q = (const u32x4*)(blobs[b] + h.off_q);
      const int nwv = (tiles_k + 7) / 8;
      if (cb == 2)
        stream_only<8, 2><<<tiles_n / 2, nwv * 64, 0, st>>>(q, tiles_k, sink);
      else
        stream_only<8, 1><<<tiles_n, nwv * 64, 0, st>>>(q, tiles_k, sink);
    };
    for (int b = 0; b < nb; ++b) launch_so(b);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
      for (int b = 0; b < nb; ++b) launch_so(b);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us2 = ms * 1e3 / (reps * nb);
    printf("         load-only twin           back-to-back %6.2f us/launch  %6.0f GB/s\n", us2, alg / us2 / 1e3);

    // timeline of one launch on a cold blob
    CK(hipMemsetAsync(probe, 0, probe_n * 8, st));
    CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe), &probe, sizeof(probe), 0, hipMemcpyHostToDevice, st));
    launch(nb / 2);
    unsigned long long* nullp = nullptr;
    CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe), &nullp, sizeof(nullp), 0, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> hp(probe_n);
    CK(hipMemcpy(hp.data(), probe, probe_n * 8, hipMemcpyDeviceToHost));
    const int nwv = (tiles_k + 7) / 8;
    const int waves = (tiles_n / cb) * 16;
    // slot 10*pass + k = clock64 at stage k (shader cycles), slot 31 = 100 MHz wall clock at entry
    const char* names[9] = {"entry", "x issued", "scales issued", "weights issued", "x staged", "tile0 done",
                            "mfma done", "barrier", "end"};
    for (int pass = 0; pass < 2; ++pass) {
      printf("    pass %d (%s instruction cache)\n", pass, pass ? "warm" : "cold?");
      for (int k = 1; k < 9; ++k) {
        std::vector<double> v;
        for (int w = 0; w < waves; ++w)
          if ((w & 15) < nwv && hp[(size_t)w * 32 + 10 * pass + k])
            v.push_back((double)(hp[(size_t)w * 32 + 10 * pass + k] - hp[(size_t)w * 32 + 10 * pass]));
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("      %-15s cycles since pass entry: min %7.0f  p10 %7.0f  med %7.0f  p90 %7.0f  max %7.0f\n", names[k],
               v[0], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
      }
    }
    for (int b = 0; b < nb; ++b) CK(hipFree(blobs[b]));
  }
  return 0;
}
