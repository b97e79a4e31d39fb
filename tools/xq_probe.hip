// xq_probe.hip — development probe for the XQ decode GEMV (not part of the product, not a test): per projection of the
// Llama-2-7B layer, back-to-back launch time over many distinct blobs (no Infinity-Cache reuse) of
//   * the round-2 kernel (gemv_xq_kernel: all tiles requested up front),
//   * the streamed kernel (gemv_xqs_kernel) over a grid of (tiles per wave, window depth),
//   * a load-only twin of the same decomposition (what the access pattern alone reaches),
// and, built with -DWOQ_XQS_STAMPS, a wall-clock timeline of one streamed launch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 -I include tools/xq_probe.hip -o tools/xq_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#ifdef WOQ_XQS_STAMPS
__device__ unsigned long long* g_xqs_probe = nullptr;
#endif
#include "../intel_extension_for_transformers_amd/csrc/woq_gemv_xq.hip"
#include "xq_r02_twin.h"

namespace woq {
int lut_args_for(uint32_t, uint32_t, LutArgs&) { return 0; }  // int4 only here (the library's lives in woq_gemv_i8.hip)
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

using woq::u32x4;

// same decomposition, loads only (rolling or not does not matter: nothing consumes)
template <int TPW, int CB>
__global__ __launch_bounds__(1024) void stream_only(const u32x4* q, int tiles_k, int base_tiles, int rem_tiles,
                                                    unsigned* sink, const unsigned char* limbs) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kt0 = wid * base_tiles + min(wid, rem_tiles);
  const int cnt = base_tiles + (wid < rem_tiles ? 1 : 0);
  u32x4 acc = {0, 0, 0, 0};
  if (limbs != nullptr) {  // + the wave's slice of the (L2-resident, shared) activation limbs: TPW x 384 bytes
    constexpr int XP = (TPW * 384 + 1023) / 1024;
    const woq::rsrc_t rl = woq::make_rsrc(limbs + (size_t)kt0 * 384, cnt * 384);
#pragma unroll
    for (int j = 0; j < XP; ++j) acc |= __builtin_amdgcn_raw_buffer_load_b128(rl, lane * 16 + j * 1024, 0, 0);
  }
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
  if ((acc.x | acc.y | acc.z | acc.w) == 0x12345u) sink[0] = 1;
}

__global__ void empty_kernel(unsigned* sink) {
  if (threadIdx.x == 12345u) sink[0] = 1;
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
__global__ void fill_f32c(float* p, size_t n, float v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_u16(unsigned short* p, size_t n, unsigned short v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

struct Shape {
  const char* name;
  int K, N, epi, norm, res, xq_out;
};

template <int TPW, int CB, int D>
static void launch_xqs(const woq::XqLaunch& a, hipStream_t st) {
  typedef woq::XqsLds<TPW, CB, 0, false, false> L;
  auto kern = woq::gemv_xqs_kernel<TPW, CB, D, 0, false, false, 0>;
  static bool attr_set = false;
  if (!attr_set) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int base = a.kt_count / a.nw, rem = a.kt_count % a.nw;
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(a.nw * 64), L::total(a.nw), st, (const u32x4*)a.q, a.scales, a.xin.limbs,
                     a.xin.u, a.tiles_k, a.kt_begin, base, rem, a.n_groups, a.tpg_shift, (const uint8_t*)a.zp, a.xin.sx,
                     a.out, a.bias, a.residual, a.eps, a.N, a.K, a.flags, a.ssq_in, a.n_ssq, a.xo, a.next_norm_w,
                     a.ssq_out, (const woq::CommDev*)nullptr, woq::LutArgs{});
}

template <int TPW, int CB>
static void launch_xqs_d(int d, const woq::XqLaunch& a, hipStream_t st) {
  switch (d) {
    case 1: launch_xqs<TPW, CB, 1>(a, st); break;
    case 2: launch_xqs<TPW, CB, 2>(a, st); break;
    case 3: launch_xqs<TPW, CB, 3>(a, st); break;
    case 4: launch_xqs<TPW, CB, 4>(a, st); break;
    default: launch_xqs<TPW, CB, TPW>(a, st); break;
  }
}

template <int CB>
static void launch_xqs_any(int tpw, int d, const woq::XqLaunch& a, hipStream_t st) {
  switch (tpw) {
    case 2: launch_xqs_d<2, CB>(d, a, st); break;
    case 3: launch_xqs_d<3, CB>(d, a, st); break;
    case 4: launch_xqs_d<4, CB>(d, a, st); break;
    case 6: launch_xqs_d<6, CB>(d, a, st); break;
    default: launch_xqs_d<8, CB>(d, a, st); break;
  }
}

template <int CB>
static void launch_so(int tpw, const u32x4* q, int grid, int nw, int tiles_k, unsigned* sink, hipStream_t st,
                      const unsigned char* limbs = nullptr) {
  const int base = tiles_k / nw, rem = tiles_k % nw;
  switch (tpw) {
    case 2: stream_only<2, CB><<<grid, nw * 64, 0, st>>>(q, tiles_k, base, rem, sink, limbs); break;
    case 3: stream_only<3, CB><<<grid, nw * 64, 0, st>>>(q, tiles_k, base, rem, sink, limbs); break;
    case 4: stream_only<4, CB><<<grid, nw * 64, 0, st>>>(q, tiles_k, base, rem, sink, limbs); break;
    case 6: stream_only<6, CB><<<grid, nw * 64, 0, st>>>(q, tiles_k, base, rem, sink, limbs); break;
    default: stream_only<8, CB><<<grid, nw * 64, 0, st>>>(q, tiles_k, base, rem, sink, limbs); break;
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const char* only = argc > 2 ? argv[2] : nullptr;
  Shape shapes[] = {{"qkv", 4096, 12288, 0, 1, 0, 0},
                    {"o", 4096, 4096, 0, 0, 1, 1},
                    {"gate_up", 4096, 22016, 1, 1, 0, 1},
                    {"down", 11008, 4096, 0, 0, 1, 1}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // XQ input (random limbs, unit factors), RMSNorm partials, residual, outputs, XQ output
  void *xin_raw, *xo_raw;
  CK(hipMalloc(&xin_raw, woq::xq_bytes(16384)));
  CK(hipMalloc(&xo_raw, woq::xq_bytes(16384)));
  float *ssq, *res, *out, *gw, *ssq_o;
  CK(hipMalloc(&ssq, 4096));
  CK(hipMalloc(&ssq_o, 4096 * 4));
  CK(hipMalloc(&res, 32768 * 4));
  CK(hipMalloc(&out, 32768 * 4));
  CK(hipMalloc(&gw, 32768 * 4));
  fill_random<<<64, 256, 0, st>>>((unsigned*)xin_raw, woq::xq_bytes(16384) / 4, 5);
  fill_f32c<<<64, 256, 0, st>>>(ssq, 1024, 1.f);
  fill_f32c<<<64, 256, 0, st>>>(res, 32768, 0.5f);
  fill_f32c<<<64, 256, 0, st>>>(gw, 32768, 1.f);
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
#ifdef WOQ_XQS_STAMPS
  unsigned long long* probe;
  const size_t probe_n = (size_t)2048 * 16 * 16;
  CK(hipMalloc(&probe, probe_n * 8));
#endif

  for (const Shape& s : shapes) {
    if (only && strcmp(only, s.name) != 0) continue;
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
    const woq::XqPtrs xin = woq::xq_carve(xin_raw, s.K);
    const woq::XqPtrs xo = woq::xq_carve(xo_raw, s.epi ? s.N / 2 : s.N);
    fill_f32c<<<64, 256, 0, st>>>(xin.u, s.K / 16, 1e-6f);
    CK(hipStreamSynchronize(st));
    const double alg = (double)s.K * s.N * 0.5 + (double)h.n_groups * s.N * 2;
    const int tiles_k = h.Kpad / 128, tiles_n = h.Npad / 16, cb = s.epi ? 2 : 1;

    auto base_args = [&](int b) {
      woq::XqLaunch a;
      a.q = blobs[b] + h.off_q;
      a.scales = blobs[b] + h.off_scale;
      a.zp = nullptr;
      a.xin = xin;
      a.tiles_k = tiles_k;
      a.K = s.K;
      a.N = s.N;
      a.n_groups = h.n_groups;
      a.tpg_shift = 0;
      a.flags = s.epi ? 2 : 0;
      a.out = s.epi ? nullptr : out;
      a.bias = nullptr;
      a.residual = s.res ? res : nullptr;
      a.eps = 1e-5f;
      a.ssq_in = s.norm ? ssq : nullptr;
      a.n_ssq = s.K / 16;
      a.xo = s.xq_out ? xo : woq::XqPtrs{nullptr, nullptr, nullptr};
      a.next_norm_w = s.xq_out && !s.epi ? gw : nullptr;
      a.ssq_out = s.xq_out && !s.epi ? ssq_o : nullptr;
      a.grid = tiles_n / cb;
      a.kt_begin = 0;
      a.kt_count = tiles_k;
      return a;
    };
    auto timeit = [&](const std::function<void(int)>& launch) {
      for (int b = 0; b < nb; ++b) launch(b);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r)
        for (int b = 0; b < nb; ++b) launch(b);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return ms * 1e3 / (reps * nb);
    };
    // the same launches captured once into a hipGraph and replayed (what the engine does)
    auto timeit_graph = [&](const std::function<void(int)>& launch) {
      hipGraph_t g;
      hipGraphExec_t ge;
      for (int b = 0; b < nb; ++b) launch(b);
      CK(hipStreamSynchronize(st));
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int b = 0; b < nb; ++b) launch(b);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
      return ms * 1e3 / (reps * nb);
    };
    printf("%-8s K=%5d N=%5d  %6.2f MB\n", s.name, s.K, s.N, alg / 1e6);
    if (getenv("XQ_PROBE_GRAPH")) {
      const int tpw = atoi(getenv("XQ_PROBE_GRAPH")) > 0 ? atoi(getenv("XQ_PROBE_GRAPH")) : 8;
      const int nw = (tiles_k + tpw - 1) / tpw;
      auto so = [&](int b) {
        if (cb == 2)
          launch_so<2>(tpw, (const u32x4*)(blobs[b] + h.off_q), tiles_n / 2, nw, tiles_k, sink, st);
        else
          launch_so<1>(tpw, (const u32x4*)(blobs[b] + h.off_q), tiles_n, nw, tiles_k, sink, st);
      };
      auto lean = [&](int b) {
        woq::XqLaunch a = base_args(b);
        a.nw = nw;
        if (cb == 2)
          launch_xqs_any<2>(tpw, 4, a, st);
        else
          launch_xqs_any<1>(tpw, 4, a, st);
      };
      auto empty = [&](int) { empty_kernel<<<tiles_n / cb, 256, 0, st>>>(sink); };
      printf("   eager / graph  empty kernel       %6.2f / %6.2f us per launch\n", timeit(empty), timeit_graph(empty));
      printf("   eager / graph  load-only twin     %6.2f / %6.2f us per launch\n", timeit(so), timeit_graph(so));
      printf("   eager / graph  lean tpw %d D 4     %6.2f / %6.2f us per launch\n", tpw, timeit(lean), timeit_graph(lean));
      for (int b = 0; b < nb; ++b) CK(hipFree(blobs[b]));
      continue;
    }
    {  // the floor: an empty kernel on the same grid (launch + boundary)
      const double us = timeit([&](int) { empty_kernel<<<tiles_n / cb, 256, 0, st>>>(sink); });
      printf("   empty kernel, same grid              %6.2f us/launch\n", us);
    }
    {  // round-2 kernel (timing twin, tools/xq_r02_twin.h) at its own geometry: 8 tiles per wave, all up front
      const int nw = (tiles_k + 7) / 8;
      const double us = timeit([&](int b) {
        woq::XqLaunch a = base_args(b);
        const size_t lds = woq::r02::xq_lds_bytes(nw, 8, cb);
        const int base = tiles_k / nw, rem = tiles_k % nw;
        if (cb == 2) {
          auto kern = woq::r02::gemv_xq_kernel<8, 2, 0, false, false>;
          static bool attr_set = false;
          if (!attr_set) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          attr_set = true;
          hipLaunchKernelGGL(kern, dim3(a.grid), dim3(nw * 64), lds, st, (const u32x4*)a.q, a.scales, a.xin.limbs,
                             a.xin.u, a.tiles_k, a.K, base, rem, a.n_groups, a.tpg_shift, (const uint8_t*)a.zp, a.xin.sx,
                             a.out, a.bias, a.residual, a.eps, a.N, a.flags, 0, a.ssq_in, a.n_ssq, a.xo, a.next_norm_w,
                             a.ssq_out);
        } else {
          auto kern = woq::r02::gemv_xq_kernel<8, 1, 0, false, false>;
          static bool attr_set = false;
          if (!attr_set) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          attr_set = true;
          hipLaunchKernelGGL(kern, dim3(a.grid), dim3(nw * 64), lds, st, (const u32x4*)a.q, a.scales, a.xin.limbs,
                             a.xin.u, a.tiles_k, a.K, base, rem, a.n_groups, a.tpg_shift, (const uint8_t*)a.zp, a.xin.sx,
                             a.out, a.bias, a.residual, a.eps, a.N, a.flags, 0, a.ssq_in, a.n_ssq, a.xo, a.next_norm_w,
                             a.ssq_out);
        }
      });
      printf("   r02 kernel (all tiles up front)      %6.2f us/launch  %6.0f GB/s\n", us, alg / us / 1e3);
    }
#ifdef WOQ_XQS_KNOBS
    {  // knock-outs of the streamed kernel at one geometry: argv[3] = tiles per wave, argv[4] = window depth
      const int tpw = argc > 3 ? atoi(argv[3]) : 8, d = argc > 4 ? atoi(argv[4]) : 4;
      const int nw = (tiles_k + tpw - 1) / tpw;
      const int exps[] = {0, 16, 32, 48, 64, 128, 192, 48 | 64, 48 | 64 | 128, 256, 256 | 16, 256 | 48, 256 | 48 | 64 | 128};
      const char* en[] = {"full", "no limb traffic", "no small loads", "no limbs, no small", "no arithmetic",
                          "no barrier/epilogue", "no arithmetic, no epilogue", "weights + waits + epilogue",
                          "weights + waits only", "no weight traffic", "no weights, no limbs", "no traffic at all",
                          "empty skeleton"};
      for (int e = 0; e < 13; ++e) {
        const double us = timeit([&](int b) {
          woq::XqLaunch a = base_args(b);
          a.nw = nw;
          a.flags |= exps[e];
          if (!a.out) a.out = out;
          if (cb == 2)
            launch_xqs_any<2>(tpw, d, a, st);
          else
            launch_xqs_any<1>(tpw, d, a, st);
        });
        printf("   tpw %d nw %2d D %d  %-28s %6.2f us/launch\n", tpw, nw, d, en[e], us);
      }
      for (int lim = 0; lim < 2; ++lim) {
        const double us = timeit([&](int b) {
          if (cb == 2)
            launch_so<2>(tpw, (const u32x4*)(blobs[b] + h.off_q), tiles_n / 2, nw, tiles_k, sink, st,
                         lim ? xin.limbs : nullptr);
          else
            launch_so<1>(tpw, (const u32x4*)(blobs[b] + h.off_q), tiles_n, nw, tiles_k, sink, st,
                         lim ? xin.limbs : nullptr);
        });
        printf("   tpw %d nw %2d      load-only twin%-14s %6.2f us/launch\n", tpw, nw, lim ? " + limb slices" : "", us);
      }
    }
    const int tpws[] = {0};
#else
    const int tpws[] = {8, 6, 4, 3, 2};
#endif
    for (int tpw : tpws) {
      if (tpw == 0) break;
      const int nw = (tiles_k + tpw - 1) / tpw;
      if (nw > 16 || nw < 1) continue;
      if (cb == 2 && tpw == 3) continue;
      {
        const double us = timeit([&](int b) {
          if (cb == 2)
            launch_so<2>(tpw, (const u32x4*)(blobs[b] + h.off_q), tiles_n / 2, nw, tiles_k, sink, st);
          else
            launch_so<1>(tpw, (const u32x4*)(blobs[b] + h.off_q), tiles_n, nw, tiles_k, sink, st);
        });
        printf("   tpw %d nw %2d  load-only twin          %6.2f us/launch  %6.0f GB/s\n", tpw, nw, us, alg / us / 1e3);
      }
      const int ds[] = {1, 2, 3, 4, 99};
      for (int d : ds) {
        if (d != 99 && d >= tpw) continue;
        const double us = timeit([&](int b) {
          woq::XqLaunch a = base_args(b);
          a.nw = nw;
          if (cb == 2)
            launch_xqs_any<2>(tpw, d, a, st);
          else
            launch_xqs_any<1>(tpw, d, a, st);
        });
        printf("   tpw %d nw %2d  streamed D=%-3s           %6.2f us/launch  %6.0f GB/s\n", tpw, nw,
               d == 99 ? "all" : std::to_string(d).c_str(), us, alg / us / 1e3);
      }
    }
    CK(hipGetLastError());
#ifdef WOQ_XQS_STAMPS
    {
      const int tpw = argc > 3 ? atoi(argv[3]) : 8, d = argc > 4 ? atoi(argv[4]) : 3;
      const int nw = (tiles_k + tpw - 1) / tpw;
      CK(hipMemsetAsync(probe, 0, probe_n * 8, st));
      CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_xqs_probe), &probe, sizeof(probe), 0, hipMemcpyHostToDevice, st));
      woq::XqLaunch a = base_args(nb / 2);
      a.nw = nw;
      if (cb == 2)
        launch_xqs_any<2>(tpw, d, a, st);
      else
        launch_xqs_any<1>(tpw, d, a, st);
      unsigned long long* nullp = nullptr;
      CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_xqs_probe), &nullp, sizeof(nullp), 0, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> hp(probe_n);
      CK(hipMemcpy(hp.data(), probe, probe_n * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      const int wgs = tiles_n / cb;
      for (int g = 0; g < wgs; ++g)
        for (int w = 0; w < nw; ++w)
          if (hp[((size_t)g * 16 + w) * 16]) t0 = std::min(t0, hp[((size_t)g * 16 + w) * 16]);
      const char* names[7] = {"entry", "loads issued", "staged", "tile 0 done", "tiles done", "barrier", "end"};
      printf("   timeline tpw %d D %d (us since the first wave's entry; 100 MHz clock)\n", tpw, d);
      for (int k = 0; k < 7; ++k) {
        std::vector<double> v;
        for (int g = 0; g < wgs; ++g)
          for (int w = 0; w < nw; ++w) {
            const unsigned long long x = hp[((size_t)g * 16 + w) * 16 + k];
            if (x) v.push_back((double)(x - t0) * 0.01);
          }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("      %-13s min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f  (%zu waves)\n", names[k], v[0],
               v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
      }
    }
#endif
    for (int b = 0; b < nb; ++b) CK(hipFree(blobs[b]));
  }
  return 0;
}
