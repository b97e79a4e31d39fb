// dma_stream_probe.hip — how fast can ONE workgroup per CU stream weights HBM -> LDS ring with LDS-DMA
// (global_load_lds_dwordx4), as a function of the number of LOADER waves, their window, and what the other waves of
// the workgroup do meanwhile (nothing / ds_read + MFMA on ring slots)? Feasibility gate of the persistent per-token
// decode engine: profiles/r02h measured ~6 B/clk per CU from a single loader wave.
//   build: hipcc --offload-arch=gfx950 -O3 tools/dma_stream_probe.hip -o tools/dma_stream_probe.bin
//   run:   tools/dma_stream_probe.bin            (prints one line per configuration)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma_4k(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// L loader waves (waves 0..L-1), C consumer waves. Workgroup b streams bytes [b * per_wg, (b + 1) * per_wg) in 4-KiB
// bursts, burst i by loader i % L, into ring slot (i % (R / 4)). D = pieces in flight per loader wave.
// Consumers (mode 1): each reads `per_wg / 1024 / C` ring tiles (ds_read_b128 + 2 MFMA + a little VALU), unsynchronised
// with the loaders: interference only.
template <int L, int D>
__global__ __launch_bounds__(1024) void stream_kernel(const uint8_t* __restrict__ src, size_t per_wg, int ring_kib,
                                                      int n_cons, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const int bursts = (int)(per_wg >> 12), ring_bursts = ring_kib >> 2;
  if (wid < L) {
    const uint8_t* p = src + (size_t)blockIdx.x * per_wg + (size_t)wid * 4096 + lane * 16;
    int slot = wid % ring_bursts;
    for (int i = wid; i < bursts; i += L) {
      lds_dma_4k(p, __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)slot * 4096u));
      p += (size_t)L * 4096;
      slot += L;
      if (slot >= ring_bursts) slot -= ring_bursts;
      wait_vmcnt<D - 4>();
    }
    wait_vmcnt<0>();
  } else if (wid < L + n_cons) {
    const int c = wid - L;
    const int tiles = (int)(per_wg >> 10);
    float acc = 0.f;
    const i32x4 izero = {0, 0, 0, 0};
    int slot = c % ring_kib;
    for (int j = c; j < tiles; j += n_cons) {
      const u32x4 wv = *(const u32x4*)(smem + (size_t)slot * 1024 + lane * 16);
      slot += n_cons;
      if (slot >= ring_kib) slot -= ring_kib;
      const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u), (int)((wv.y << 4) & 0xf0f0f0f0u),
                        (int)(wv.y & 0xf0f0f0f0u)};
      const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u), (int)((wv.w << 4) & 0xf0f0f0f0u),
                        (int)(wv.w & 0xf0f0f0f0u)};
      const i32x4 a0 = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
      const i32x4 d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, izero, 0, 0, 0);
      const i32x4 d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b1, izero, 0, 0, 0);
      acc = fmaf((float)(d0.x + d0.y), 1.0f / 65536.f, fmaf((float)(d1.z + d1.w), 0.5f, acc));
    }
    if (acc == 1234.5f) sink[blockIdx.x] = acc;
  }
}

// reference: every wave loads to registers (the twin the bench calls the "ceiling"), 16 waves x 4 x 16 B in flight
__global__ __launch_bounds__(1024) void reg_stream_kernel(const uint8_t* __restrict__ src, size_t per_wg,
                                                          float* __restrict__ sink) {
  const int tid = threadIdx.x;
  const u32x4* p = (const u32x4*)(src + (size_t)blockIdx.x * per_wg) + tid;
  const int n = (int)(per_wg >> 14);  // 16 KiB per workgroup per iteration
  u32x4 acc = {0, 0, 0, 0};
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    const u32x4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1024),
                c = __builtin_nontemporal_load(p + 2048), d = __builtin_nontemporal_load(p + 3072);
    p += 4096;
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n; ++i) {
    acc ^= __builtin_nontemporal_load(p);
    p += 1024;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1.f;
}

template <typename F>
static double time_graph(F&& launch, int reps, hipStream_t st) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < reps; ++r) launch(r);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return best / reps * 1e3;  // us per launch
}

int main() {
  const size_t total = (size_t)3 << 30;  // 3 GiB: far beyond the 256 MiB MALL; launches rotate through it
  uint8_t* buf;
  float* sink;
  CK(hipMalloc(&buf, total));
  CK(hipMemset(buf, 0x5a, total));
  CK(hipMalloc(&sink, 4096));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int G = 256, reps = 24;
  const size_t sizes[] = {(size_t)24 << 20, (size_t)96 << 20};  // a qkv projection; a whole layer
  for (size_t bytes : sizes) {
    const size_t per_wg = bytes / G;
    const int n_off = (int)(total / bytes);
    {
      auto f = [&](int r) {
        hipLaunchKernelGGL(reg_stream_kernel, dim3(G), dim3(1024), 0, st, buf + (size_t)(r % n_off) * bytes, per_wg, sink);
      };
      const double us = time_graph(f, reps, st);
      printf("bytes %zu MiB  regs 16 waves            : %7.2f us  %6.2f TB/s\n", bytes >> 20, us, bytes / us * 1e-6);
    }
#define RUN(L, D, NC)                                                                                                 \
  {                                                                                                                   \
    auto kern = stream_kernel<L, D>;                                                                                  \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
    auto f = [&](int r) {                                                                                             \
      hipLaunchKernelGGL(kern, dim3(G), dim3(1024), 128 * 1024, st, buf + (size_t)(r % n_off) * bytes, per_wg, 128,   \
                         NC, sink);                                                                                   \
    };                                                                                                                \
    const double us = time_graph(f, reps, st);                                                                        \
    printf("bytes %zu MiB  dma L=%d D=%2d consumers=%2d : %7.2f us  %6.2f TB/s  %5.2f B/clk/CU\n", bytes >> 20, L, D, \
           NC, us, bytes / us * 1e-6, bytes / us * 1e-6 / 256 / 2.4e-3);                                              \
  }
    RUN(1, 32, 0)
    RUN(1, 60, 0)
    RUN(2, 16, 0)
    RUN(2, 32, 0)
    RUN(2, 60, 0)
    RUN(4, 8, 0)
    RUN(4, 16, 0)
    RUN(4, 32, 0)
    RUN(8, 8, 0)
    RUN(8, 16, 0)
    RUN(2, 32, 12)
    RUN(4, 16, 12)
    RUN(4, 32, 12)
    RUN(4, 16, 8)
    RUN(8, 16, 8)
  }
  // one launch streaming a whole token's weights (3.4 GiB would not fit: 2.9 GiB), to see the steady state
  {
    const size_t bytes = (size_t)2944 << 20;
    const size_t per_wg = bytes / G;
    auto kern = stream_kernel<4, 16>;
    auto f = [&](int) { hipLaunchKernelGGL(kern, dim3(G), dim3(1024), 128 * 1024, st, buf, per_wg, 128, 12, sink); };
    const double us = time_graph(f, 2, st);
    printf("bytes %zu MiB  dma L=4 D=16 consumers=12 one launch: %7.2f us  %6.2f TB/s\n", bytes >> 20, us, bytes / us * 1e-6);
    auto kern2 = stream_kernel<2, 32>;
    CK(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto f2 = [&](int) { hipLaunchKernelGGL(kern2, dim3(G), dim3(1024), 128 * 1024, st, buf, per_wg, 128, 12, sink); };
    const double us2 = time_graph(f2, 2, st);
    printf("bytes %zu MiB  dma L=2 D=32 consumers=12 one launch: %7.2f us  %6.2f TB/s\n", bytes >> 20, us2, bytes / us2 * 1e-6);
    auto f3 = [&](int) { hipLaunchKernelGGL(reg_stream_kernel, dim3(G), dim3(1024), 0, st, buf, per_wg, sink); };
    const double us3 = time_graph(f3, 2, st);
    printf("bytes %zu MiB  regs one launch: %7.2f us  %6.2f TB/s\n", bytes >> 20, us3, bytes / us3 * 1e-6);
  }
  return 0;
}
