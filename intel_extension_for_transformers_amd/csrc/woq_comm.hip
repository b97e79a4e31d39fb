// woq_comm.hip — device-side tensor-parallel exchange for batch-1 decode over xGMI: a one-shot all-reduce of the
// row-parallel partial sums and the greedy-token exchange, both plain kernels on the engine's stream, so a whole TP
// token step replays as ONE hipGraph with no host in the loop.
//
// Why not RCCL here. The only multi-device precedent in the reference is DeepSpeed AutoTP over oneCCL / HCCL on fp
// models (neural_chat/models/model_utils.py:238-311); SURVEY.md §8(e) specifies the exchange for this path: one
// all-reduce (sum) of [1, hidden] fp32 after o_proj and one after down_proj — 160 per token for Llama-2-70B, 32 KiB
// each. At that size a collective is pure latency: a host-issued RCCL call costs ~20 us and cannot ride in the
// engine's captured graph; ring steps add 2 (W - 1) hops. MI355X's xGMI is a full mesh of point-to-point links, so the
// natural form is ONE hop: every rank stores its vector straight into every peer's inbox and then sums the W vectors
// that arrive in its own. RCCL stays the transport of the prompt pass (bandwidth-bound [rows, hidden] sums) and of
// process-group setup (runtime/comm.py).
//
// Protocol (the "LL" idea: data and flag travel in one store, so no fence and no flag can overtake its data).
//  * granule = one naturally aligned 8-byte word {fp32 payload, 32-bit tag}, written by ONE system-scope store and read
//    by ONE system-scope load; tag = the collective's sequence number (never 0; inboxes start zeroed).
//  * inbox (per rank, in that rank's HBM, IPC-mapped by every peer): [2 buffers][W senders][max_elems] granules.
//    Collective s uses buffer s & 1. A rank can only push collective s + 2 after it finished s + 1, which needed the
//    peer's s + 1 contribution, which the peer sent only after it had consumed s: two buffers are enough.
//  * the sequence number lives on the device (ctl[0]) and is advanced by the LAST workgroup of each collective kernel
//    (arrival ticket), so a captured graph replays with fresh tags; every rank runs the same chain of collectives.
//  * sums run in rank order 0..W-1 on every rank: all ranks hold bit-identical results.
//  * every spin is bounded (wall clock): on a timeout the kernel records status != 0 and finishes, it never hangs the GPU.
// The inbox is allocated uncached / fine-grained (hipExtMallocWithFlags) because peers write it over the fabric behind
// the local L2's back; loads and stores of granules are system scope (sc0 sc1).
#include <cstring>
#include <vector>

#include "woq_device.h"
#include "woq_launch.h"

#define WOQ_COMM_MAX_WORLD 8

namespace woq {

struct CommDev {
  uint64_t* peer[WOQ_COMM_MAX_WORLD];  // inbox base of every rank (peer[rank] = the local one)
  uint32_t* ctl;                       // [0] sequence number, [1] arrival ticket, [2] status (0 ok, else timeouts seen)
  int rank, world;
  uint32_t max_elems;
  uint32_t timeout_ticks;              // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ size_t ar_slot(const CommDev& c, int buf, int sender, uint32_t i) {
  return ((size_t)buf * c.world + sender) * c.max_elems + i;
}
__device__ __forceinline__ size_t am_slot(const CommDev& c, int buf, int sender, int j) {
  return (size_t)2 * c.world * c.max_elems + ((size_t)buf * c.world + sender) * 2 + j;
}
__device__ __forceinline__ void push(uint64_t* p, uint32_t payload, uint32_t tag) {
  __hip_atomic_store(p, ((uint64_t)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spin until the granule carries `tag`; false on timeout
__device__ __forceinline__ bool pull(const uint64_t* p, uint32_t tag, uint64_t t0, uint32_t limit, uint32_t& payload) {
  for (int spins = 1;; ++spins) {
    const uint64_t g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((uint32_t)(g >> 32) == tag) {
      payload = (uint32_t)g;
      return true;
    }
    if ((spins & 31) == 0 && wall_clock64() - t0 > limit) {
      payload = 0;
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
// 0 is never a live tag; on the wrap 0xffffffff (odd) -> 2 (even), so consecutive collectives still alternate buffers
__device__ __forceinline__ uint32_t next_seq(uint32_t s) { return s + 1 == 0 ? 2 : s + 1; }

// in-place sum over ranks of buf[0..n). grid = ceil(n / 1024) x 256 threads, 4 elements per thread (stride 256: every
// store instruction of a wave covers 512 contiguous bytes of one peer's inbox).
__global__ __launch_bounds__(256) void allreduce_ll_kernel(CommDev c, float* __restrict__ buf, uint32_t n) {
  // the grid size is read HERE and kept in a register: left to itself hipcc loads it from the kernel-argument segment
  // at the tail and recycles the segment pointer right behind the load (the pattern of DESIGN.md §3.3's trap)
  uint32_t nblk = gridDim.x;
  asm volatile("" : "+s"(nblk));
  const uint32_t seq = c.ctl[0];
  const int b = (int)(seq & 1u);
  const uint32_t i0 = blockIdx.x * 1024u + threadIdx.x;
  float mine[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t i = i0 + j * 256u;
    mine[j] = i < n ? buf[i] : 0.f;
  }
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i = i0 + j * 256u;
      if (i < n) push(c.peer[r] + ar_slot(c, b, c.rank, i), __float_as_uint(mine[j]), seq);
    }
  }
  const uint64_t t0 = wall_clock64();
  bool ok = true;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < c.world; ++r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i = i0 + j * 256u;
      if (i >= n) continue;
      float v = mine[j];
      if (r != c.rank) {
        uint32_t bits;
        ok &= pull(c.peer[c.rank] + ar_slot(c, b, r, i), seq, t0, c.timeout_ticks, bits);
        v = __uint_as_float(bits);
      }
      acc[j] += v;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t i = i0 + j * 256u;
    if (i < n) buf[i] = acc[j];
  }
  if (!ok) atomicOr(&c.ctl[2], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    // every workgroup read ctl[0] before it got here, so the last one to arrive may advance it
    if (atomicAdd(&c.ctl[1], 1u) == nblk - 1) {
      atomicExch(&c.ctl[1], 0u);
      atomicExch(&c.ctl[0], next_seq(seq));
    }
  }
}

// greedy token of a vocab-sharded lm_head: local argmax over the lm_head launch's per-workgroup (max, index) pairs,
// one (max, global index) pair per rank exchanged through the inboxes, winner = highest value, lowest global index
// on ties (a single-device argmax over the gathered row). Writes token and advances pos. One workgroup.
__global__ __launch_bounds__(1024) void tp_greedy_kernel(CommDev c, const float* __restrict__ pmax,
                                                         const int32_t* __restrict__ pidx, int n, int vocab_offset,
                                                         int32_t* __restrict__ token, int32_t* __restrict__ pos,
                                                         int32_t* __restrict__ log) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = tid; i < n; i += 1024) {
    const float v = pmax[i];
    const int ci = pidx[i];
    if (v > best || (v == best && ci < idx)) {
      best = v;
      idx = ci;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) {
      best = ov;
      idx = oi;
    }
  }
  if (lane == 0) {
    bv[wid] = best;
    bi[wid] = idx;
  }
  __syncthreads();
  if (tid != 0) return;
  for (int w = 1; w < 16; ++w)
    if (bv[w] > best || (bv[w] == best && bi[w] < idx)) {
      best = bv[w];
      idx = bi[w];
    }
  const uint32_t seq = c.ctl[0];
  const int b = (int)(seq & 1u);
  const int gidx = idx == 0x7fffffff ? idx : idx + vocab_offset;
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) continue;
    push(c.peer[r] + am_slot(c, b, c.rank, 0), __float_as_uint(best), seq);
    push(c.peer[r] + am_slot(c, b, c.rank, 1), (uint32_t)gidx, seq);
  }
  const uint64_t t0 = wall_clock64();
  bool ok = true;
  float top = best;
  int top_i = gidx;
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) continue;
    uint32_t vb, ib;
    ok &= pull(c.peer[c.rank] + am_slot(c, b, r, 0), seq, t0, c.timeout_ticks, vb);
    ok &= pull(c.peer[c.rank] + am_slot(c, b, r, 1), seq, t0, c.timeout_ticks, ib);
    const float v = __uint_as_float(vb);
    if (v > top || (v == top && (int)ib < top_i)) {
      top = v;
      top_i = (int)ib;
    }
  }
  token[0] = top_i;
  if (log != nullptr) log[pos[0]] = top_i;
  pos[0] = pos[0] + 1;
  if (!ok) atomicOr(&c.ctl[2], 2u);
  atomicExch(&c.ctl[0], next_seq(seq));
}

}  // namespace woq

struct woq_comm {
  woq::CommDev dev;
  size_t inbox_bytes = 0;
  void* opened[WOQ_COMM_MAX_WORLD] = {};
  hipIpcMemHandle_t handle;
  bool have_handle = false, connected = false;
};

size_t woq_comm_max_elems(woq_comm* c) { return c ? c->dev.max_elems : 0; }

int woq_comm_launch_allreduce(woq_comm* c, float* buf, size_t n, hipStream_t st) {
  if (!c || !c->connected) return woq::fail("QBits: tensor-parallel communicator is not connected");
  if (n > c->dev.max_elems) return woq::fail("QBits: all-reduce larger than the communicator's inbox");
  if (c->dev.world == 1 || n == 0) return 0;
  hipLaunchKernelGGL(woq::allreduce_ll_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, c->dev, buf,
                     (uint32_t)n);
  return 0;
}

int woq_comm_launch_greedy(woq_comm* c, const float* pmax, const int32_t* pidx, int n, int vocab_offset,
                           int32_t* token, int32_t* pos, int32_t* log, hipStream_t st) {
  if (!c || !c->connected) return woq::fail("QBits: tensor-parallel communicator is not connected");
  hipLaunchKernelGGL(woq::tp_greedy_kernel, dim3(1), dim3(1024), 0, st, c->dev, pmax, pidx, n, vocab_offset, token,
                     pos, log);
  return 0;
}

extern "C" {

int woq_comm_create(int rank, int world, size_t max_elems, woq_comm** out) {
  WOQ_TRY
  WOQ_CHECK(out && world >= 1 && world <= WOQ_COMM_MAX_WORLD && rank >= 0 && rank < world,
            "QBits: tensor-parallel world size must be 1..8");
  WOQ_CHECK(max_elems >= 1 && max_elems < (1u << 28), "QBits: bad communicator size");
  woq_comm* c = new woq_comm();
  c->dev.rank = rank;
  c->dev.world = world;
  c->dev.max_elems = (uint32_t)max_elems;
  c->dev.timeout_ticks = 200000000u;  // 2 s at 100 MHz
  c->inbox_bytes = ((size_t)2 * world * (max_elems + 2)) * sizeof(uint64_t);
  void* inbox = nullptr;
  // peers write this memory over xGMI: it must not be cached by the local L2 (uncached > fine-grained > plain)
  hipError_t e = hipExtMallocWithFlags(&inbox, c->inbox_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&inbox, c->inbox_bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    return woq::fail(std::string("QBits: cannot allocate the tensor-parallel inbox: ") + hipGetErrorString(e));
  }
  WOQ_HIP(hipMemset(inbox, 0, c->inbox_bytes));
  WOQ_HIP(hipMalloc((void**)&c->dev.ctl, 64));
  const uint32_t init[4] = {1u, 0u, 0u, 0u};
  WOQ_HIP(hipMemcpy(c->dev.ctl, init, sizeof(init), hipMemcpyHostToDevice));
  WOQ_HIP(hipDeviceSynchronize());
  for (int r = 0; r < WOQ_COMM_MAX_WORLD; ++r) c->dev.peer[r] = nullptr;
  c->dev.peer[rank] = (uint64_t*)inbox;
  c->connected = world == 1;
  *out = c;
  WOQ_END
}

int woq_comm_handle(woq_comm* c, void* handle_out, size_t bytes) {
  WOQ_TRY
  WOQ_CHECK(c && handle_out && bytes >= sizeof(hipIpcMemHandle_t), "QBits: handle buffer must hold 64 bytes");
  if (!c->have_handle) {
    WOQ_HIP(hipIpcGetMemHandle(&c->handle, c->dev.peer[c->dev.rank]));
    c->have_handle = true;
  }
  memcpy(handle_out, &c->handle, sizeof(hipIpcMemHandle_t));
  WOQ_END
}

int woq_comm_connect(woq_comm* c, const void* handles, const int* peer_devices) {
  WOQ_TRY
  WOQ_CHECK(c && handles, "QBits: null communicator");
  int me = 0;
  WOQ_HIP(hipGetDevice(&me));
  for (int r = 0; r < c->dev.world; ++r) {
    if (r == c->dev.rank) continue;
    if (peer_devices && peer_devices[r] != me) {
      int can = 0;
      WOQ_HIP(hipDeviceCanAccessPeer(&can, me, peer_devices[r]));
      WOQ_CHECK(can, "QBits: no peer access between the tensor-parallel GPUs");
      hipError_t pe = hipDeviceEnablePeerAccess(peer_devices[r], 0);
      if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
        return woq::fail(std::string("QBits: hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
      (void)hipGetLastError();
    }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)r * sizeof(hipIpcMemHandle_t), sizeof(h));
    void* p = nullptr;
    WOQ_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    c->opened[r] = p;
    c->dev.peer[r] = (uint64_t*)p;
  }
  c->connected = true;
  WOQ_END
}

int woq_comm_allreduce_f32(woq_comm* c, float* buf_dev, size_t n, void* stream) {
  WOQ_TRY
  int rc = woq_comm_launch_allreduce(c, buf_dev, n, (hipStream_t)stream);
  if (rc) return rc;
  WOQ_HIP(hipGetLastError());
  WOQ_END
}

int woq_comm_status(woq_comm* c, void* stream, int* status_out) {
  WOQ_TRY
  WOQ_CHECK(c && status_out, "QBits: null communicator");
  uint32_t host[4] = {0, 0, 0, 0};
  WOQ_HIP(hipStreamSynchronize((hipStream_t)stream));
  WOQ_HIP(hipMemcpy(host, c->dev.ctl, sizeof(host), hipMemcpyDeviceToHost));
  *status_out = (int)host[2];
  WOQ_END
}

int woq_comm_set_timeout_ms(woq_comm* c, int ms) {
  WOQ_TRY
  WOQ_CHECK(c && ms >= 1 && ms <= 20000, "QBits: communicator timeout must be 1..20000 ms");
  c->dev.timeout_ticks = (uint32_t)ms * 100000u;
  WOQ_END
}

void woq_comm_destroy(woq_comm* c) {
  if (!c) return;
  for (int r = 0; r < WOQ_COMM_MAX_WORLD; ++r)
    if (c->opened[r]) hipIpcCloseMemHandle(c->opened[r]);
  if (c->dev.peer[c->dev.rank]) hipFree(c->dev.peer[c->dev.rank]);
  if (c->dev.ctl) hipFree(c->dev.ctl);
  delete c;
}

}  // extern "C"
