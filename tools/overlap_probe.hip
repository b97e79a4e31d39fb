// overlap_probe.hip — can consecutive DEPENDENT decode launches overlap their heads and tails on this stack?
//
// The batch-1 decode step is 128 launches per token, each ~1.5 us of kernel boundary + ~1 us until its first bytes arrive
// + a tail, during which HBM idles. A launch's weight requests do not depend on its predecessor — only
// its activation vector does. If launch k + 1 could START while launch k is still running (its weight tiles in flight,
// its waves polling for k's outputs), the boundary, the head and most of the tail would hide under the neighbours'
// streams. Two ways to ask the hardware for that without a persistent kernel:
//   (A) hipExtLaunchKernel(..., hipExtAnyOrderLaunch): the AQL packet without its barrier bit, same queue — dispatch
//       order is kept (the packet processor launches a packet's workgroups in order), completion order is not;
//   (B) two (or three) streams taking the launches in rotation, no events between them: stream order still serialises
//       k and k + 2, the in-kernel wait orders k and k + 1.
// This probe measures both against plain serial launches on a chain of load-only kernels shaped like the Llama-2-7B
// layer's four GEMVs (same grids, waves, bytes; every buffer distinct, 3.3 GB in rotation: cold), with the dependency
// modelled as an all-to-all arrival counter per launch (every workgroup of launch k adds 1 when done; every wave of
// launch k + 1 waits — AFTER issuing its loads — until the counter reaches k's grid size).
// Build: hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe.bin
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

// one wave = 8 KiB of its workgroup's contiguous range. Dependency protocol = the engine's own (woq_xq.h XqPub): producer
// workgroup b publishes ONE flag word (write-through agent-scope store, the round number) when it is done; a consumer
// wave waits for the flags of ITS K slice (the producer's n_prev flags cut evenly over the consumer's waves, one flag
// per lane and load) — no atomics, no all-to-one counter. wait = 0: no dependency.
// poll: 0 every wave polls the flags of its slice | 1 wave 0 polls ALL flags, the others wait at the workgroup barrier
// publish: 0 nothing is stored (calibration against the engine's load-only twins) | 1 one flag word per workgroup
__global__ __launch_bounds__(1024) void link_kernel(const u32x4* __restrict__ w, const unsigned int* prev, int n_prev,
                                                    unsigned int target, unsigned int* mine, unsigned int round_no,
                                                    int wait, int work_clocks, int* status, unsigned int* sink, int poll,
                                                    int sleep_n, int publish, unsigned long long* stamps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = wall_clock64();
  const u32x4* p = w + ((size_t)blockIdx.x * nw + wid) * 8 * 64 + lane;
  u32x4 v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = __builtin_nontemporal_load(p + t * 64);
  if (wait) {
    const int per = poll == 1 ? n_prev : (n_prev + nw - 1) / nw, f0 = poll == 1 ? 0 : wid * per;
    const int cnt = max(0, min(per, n_prev - f0));
    if (poll == 0 || wid == 0) {
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        bool good = true;
        for (int j = 0; j < cnt; j += 64)
          if (lane + j < cnt) {
            const unsigned int c = __hip_atomic_load(prev + f0 + j + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            good = good && (int)(c - target) >= 0;
          }
        if (__all(good)) break;
        if (wall_clock64() - t0 > 100000ull) {  // 1 ms at 100 MHz: give up, say so
          if (lane == 0) atomicOr(status, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        for (int q = 1; q < sleep_n; ++q) __builtin_amdgcn_s_sleep(1);
      }
    }
    if (poll == 1) __syncthreads();
  }
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[1] = wall_clock64();
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; ++t) acc |= v[t];
  if (work_clocks > 0) {  // stand-in for the arithmetic tail of a GEMV wave
    const unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long)work_clocks) __builtin_amdgcn_s_sleep(1);
  }
  if ((acc.x & acc.y & acc.z & acc.w) == 0x9e3779b9u) sink[0] = 1;
  __syncthreads();
  if (publish && threadIdx.x == 0)
    __hip_atomic_store(mine + blockIdx.x, round_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[2] = wall_clock64();
}

struct Shape {
  int grid, waves;
};

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 32;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int work = argc > 3 ? atoi(argv[3]) : 0;
  const int poll = argc > 4 ? atoi(argv[4]) : 0;
  const int sleep_n = argc > 5 ? atoi(argv[5]) : 8;
  const Shape shapes[4] = {{768, 4}, {256, 4}, {688, 8}, {256, 11}};  // qkv, o, gate/up pairs, down (8 KiB per wave)
  const int n = layers * 4;
  std::vector<u32x4*> bufs(n);
  std::vector<size_t> bytes(n);
  double total_bytes = 0;
  for (int i = 0; i < n; ++i) {
    const Shape& s = shapes[i & 3];
    bytes[i] = (size_t)s.grid * s.waves * 8192;
    CK(hipMalloc((void**)&bufs[i], bytes[i]));
    CK(hipMemset(bufs[i], 1, bytes[i]));
    total_bytes += (double)bytes[i];
  }
  unsigned int* ctr;
  CK(hipMalloc((void**)&ctr, (size_t)(n + 1) * 4096));  // 1024 flag words per launch
  int* status;
  CK(hipMalloc((void**)&status, 8));
  unsigned int* sink = (unsigned int*)(status + 1);
  hipStream_t st[3];
  for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long* stamps;
  CK(hipMalloc((void**)&stamps, (size_t)n * 4 * 8));
  printf("chain of %d load-only launches (%d layers x qkv/o/gate_up/down shapes), %.1f MB per pass, tail work %d clocks, "
         "poll form %d, sleep %d x 64 clocks\n", n, layers, total_bytes / 1e6, work, poll, sleep_n);
  // modes: 0 serial, no waits | 1 serial + waits (always satisfied) | 2 any-order flag, one stream | 3 two streams in
  // rotation | 4 three streams in rotation
  const char* names[6] = {"serial launches, no wait, NO flag store (= engine twins)", "serial launches, no in-kernel wait", "serial launches + in-kernel wait",
                          "hipExtAnyOrderLaunch, one stream + in-kernel wait", "two streams in rotation + in-kernel wait",
                          "three streams in rotation + in-kernel wait"};
  for (int mode = -1; mode < 5; ++mode) {
    CK(hipMemset(ctr, 0, (size_t)(n + 1) * 4096));
    CK(hipMemset(status, 0, 8));
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    unsigned int round = 0;
    for (int r = 0; r <= reps; ++r) {
      CK(hipEventRecord(e0, st[0]));
      for (int i = 0; i < n; ++i) {
        const Shape& s = shapes[i & 3];
        // launch i waits for launch i - 1 of the SAME pass (round r + 1); launch 0 for the last launch of the previous pass
        const int ip = i == 0 ? n - 1 : i - 1;
        unsigned int* prev = ctr + (size_t)ip * 1024;
        const int n_prev = shapes[ip & 3].grid;
        const unsigned int target = i == 0 ? round : round + 1;
        unsigned int* mine = ctr + (size_t)i * 1024;
        const int wait = mode >= 1 ? 1 : 0;
        hipStream_t s_use = mode == 3 ? st[i & 1] : (mode == 4 ? st[i % 3] : st[0]);
        if (mode == 2) {
          hipExtLaunchKernelGGL(link_kernel, dim3(s.grid), dim3(s.waves * 64), 0, s_use, nullptr, nullptr,
                                hipExtAnyOrderLaunch, (const u32x4*)bufs[i], (const unsigned int*)prev, n_prev, target, mine,
                                round + 1, wait, work, status, sink, poll, sleep_n, 1, stamps + (size_t)i * 4);
        } else {
          hipLaunchKernelGGL(link_kernel, dim3(s.grid), dim3(s.waves * 64), 0, s_use, (const u32x4*)bufs[i],
                             (const unsigned int*)prev, n_prev, target, mine, round + 1, wait, work, status, sink, poll,
                             sleep_n, mode >= 0 ? 1 : 0, stamps + (size_t)i * 4);
        }
      }
      ++round;
      if (mode >= 3) {  // the timing stream waits for the others (outside the measured chain's steady state)
        for (int k = 1; k < (mode == 3 ? 2 : 3); ++k) {
          hipEvent_t ej;
          CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
          CK(hipEventRecord(ej, st[k]));
          CK(hipStreamWaitEvent(st[0], ej, 0));
          CK(hipEventDestroy(ej));
        }
      }
      CK(hipEventRecord(e1, st[0]));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) best = ms;
    }
    int hstat = 0;
    CK(hipMemcpy(&hstat, status, 4, hipMemcpyDeviceToHost));
    if (mode == 1 || mode == 2) {  // timeline of launches 40..47 of the last pass, workgroup 0: start | flags seen | end
      std::vector<unsigned long long> h((size_t)n * 4);
      CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      const double t0 = (double)h[40 * 4];
      printf("   timeline (us from launch 40's start; wall clock 100 MHz):");
      for (int i = 40; i < 48; ++i)
        printf("  [%d] %.2f %.2f %.2f", i, (h[i * 4] - t0) / 100.0, (h[i * 4 + 1] - t0) / 100.0, (h[i * 4 + 2] - t0) / 100.0);
      printf("\n");
    }
    printf("mode %d  %-52s  %8.1f us per pass  %6.2f us per launch  %6.2f TB/s  gave-up-waiting=%d\n", mode, names[mode + 1],
           best * 1e3, best * 1e3 / n, total_bytes / (best * 1e-3) / 1e12, hstat);
  }
  return 0;
}
