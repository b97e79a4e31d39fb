// kernarg_probe.hip — how long does a wave wait for a kernel argument that is NOT preloaded into SGPRs?
// Each wave's lane 0 times (s_memtime) an s_load of an argument dword at offset 0x90 of the kernarg segment and of a
// dword in device memory; launched eagerly and from a graph. hipcc -O3 --offload-arch=gfx950 tools/kernarg_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Pad {
  int v[40];
};

__global__ void probe(unsigned long long* out, const int* dev_word, Pad pad) {
  const int* ka = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned long long t0, t1, t2;
  int a, b;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  asm volatile("s_load_dword %0, %1, 0x90\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(ka) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b) : "s"(dev_word) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[w * 2] = (t1 - t0) + (a == 123456789 ? 1 : 0);
    out[w * 2 + 1] = (t2 - t1) + (b == 123456789 ? 1 : 0);
  }
}

static void report(const char* name, unsigned long long* dev, int waves) {
  std::vector<unsigned long long> h(waves * 2);
  CK(hipMemcpy(h.data(), dev, waves * 16, hipMemcpyDeviceToHost));
  for (int k = 0; k < 2; ++k) {
    std::vector<unsigned long long> v;
    for (int w = 0; w < waves; ++w) v.push_back(h[w * 2 + k]);
    std::sort(v.begin(), v.end());
    printf("%-28s %-16s cycles min %5llu  p10 %5llu  med %5llu  p90 %5llu  max %5llu\n", name,
           k ? "device-memory word" : "kernarg word", v[0], v[waves / 10], v[waves / 2], v[waves * 9 / 10], v.back());
  }
}

int main() {
  const int grid = 768, block = 256, waves = grid * block / 64;
  unsigned long long* out;
  int* word;
  CK(hipMalloc(&out, waves * 16));
  CK(hipMalloc(&word, 64 * 1024 * 1024));
  CK(hipMemset(word, 0, 64 * 1024 * 1024));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  Pad pad;
  for (int i = 0; i < 40; ++i) pad.v[i] = i;
  for (int rep = 0; rep < 3; ++rep) {
    probe<<<grid, block, 0, st>>>(out, word + rep * 1024 * 1024, pad);
    CK(hipStreamSynchronize(st));
    report("eager launch", out, waves);
  }
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  probe<<<grid, block, 0, st>>>(out, word + 8 * 1024 * 1024, pad);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    report("graph replay", out, waves);
  }
  return 0;
}
