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

#include "woq_comm_dev.h"
#include "woq_xq.h"

namespace woq {

// in-place sum over ranks of buf[0..n). grid = ceil(n / 1024) x 256 threads, 4 elements per thread (stride 256: every
// store instruction of a wave covers 512 contiguous bytes of one peer's inbox).
//   PUSHED: this rank's contribution is already on its way — the kernel that produced `buf` stored it into the peers'
//           inboxes from its own epilogue (woq_gemv_xqs.h: the 16 epilogue lanes of every column strip), so the fabric
//           flight runs under the kernel boundary instead of after it; this kernel only pulls and sums.
//   XQ:     the summed vector also leaves as the next GEMV's XQ input (woq_xq.h): times `norm_w`, per-block sums of
//           squares of the raw sums in `ssq_out` — thread t holds 16 consecutive elements per DPP row for each j.
template <bool PUSHED, bool XQ>
__global__ __launch_bounds__(256) void allreduce_ll_kernel(CommDev c, float* __restrict__ buf, uint32_t n,
                                                           const float* __restrict__ norm_w, XqPtrs xo,
                                                           float* __restrict__ ssq_out) {
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
  if constexpr (!PUSHED) {
    for (int r = 0; r < c.world; ++r) {
      if (r == c.rank) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t i = i0 + j * 256u;
        if (i < n) push(c.peer[r] + ar_slot(c, b, c.rank, i), __float_as_uint(mine[j]), seq);
      }
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
    if constexpr (XQ) {  // n is a multiple of 16 (checked by the launcher): whole DPP rows are live or dead together
      if (i < n) {
        if (ssq_out != nullptr) {
          const float ss = row16_sum(acc[j] * acc[j]);
          if ((i & 15u) == 0) ssq_out[i >> 4] = ss;
        }
        xq_emit16(norm_w != nullptr ? acc[j] * norm_w[i] : acc[j], xo, (int)(i >> 4), (int)(i & 15u));
      }
    }
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
  woq::CommDev* dev_copy = nullptr;  // the same record in device memory, for kernels that take it by pointer
  size_t inbox_bytes = 0;
  void* opened[WOQ_COMM_MAX_WORLD] = {};
  hipIpcMemHandle_t handle;
  bool have_handle = false, connected = false;
};

size_t woq_comm_max_elems(woq_comm* c) { return c ? c->dev.max_elems : 0; }

static const woq::XqPtrs kCommNoXq = {nullptr, nullptr, nullptr};

int woq_comm_launch_allreduce(woq_comm* c, float* buf, size_t n, hipStream_t st) {
  if (!c || !c->connected) return woq::fail("QBits: tensor-parallel communicator is not connected");
  if (n > c->dev.max_elems) return woq::fail("QBits: all-reduce larger than the communicator's inbox");
  if (c->dev.world == 1 || n == 0) return 0;
  hipLaunchKernelGGL((woq::allreduce_ll_kernel<false, false>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st,
                     c->dev, buf, (uint32_t)n, (const float*)nullptr, kCommNoXq, (float*)nullptr);
  return 0;
}

// the engine's form: `pushed` = the producing GEMV already stored this rank's contribution into the peers' inboxes
// (woq_comm_dev_ptr), `xo` (optional) = the summed vector also as an XQ vector (times norm_w, sums of squares in ssq_out)
int woq_comm_launch_allreduce_ex(woq_comm* c, float* buf, size_t n, int pushed, const float* norm_w,
                                 const woq::XqPtrs& xo, float* ssq_out, hipStream_t st) {
  if (!c || !c->connected) return woq::fail("QBits: tensor-parallel communicator is not connected");
  if (n > c->dev.max_elems) return woq::fail("QBits: all-reduce larger than the communicator's inbox");
  if (n == 0) return 0;
  const bool xq = xo.limbs != nullptr;
  if (xq && (n & 15) != 0) return woq::fail("QBits: an XQ-emitting all-reduce needs a multiple of 16 elements");
  const dim3 grid((unsigned)((n + 1023) / 1024));
#define WOQ_AR_CASE(P, X)                                                                                         \
  if ((pushed != 0) == P && xq == X) {                                                                             \
    hipLaunchKernelGGL((woq::allreduce_ll_kernel<P, X>), grid, dim3(256), 0, st, c->dev, buf, (uint32_t)n, norm_w, \
                       xo, ssq_out);                                                                                \
    return 0;                                                                                                      \
  }
  WOQ_AR_CASE(false, false)
  WOQ_AR_CASE(false, true)
  WOQ_AR_CASE(true, false)
  WOQ_AR_CASE(true, true)
#undef WOQ_AR_CASE
  return 0;
}

// device-resident copy of the communicator record (peer inboxes, sequence word, rank / world), valid once connected
const woq::CommDev* woq_comm_dev_ptr(woq_comm* c) { return c && c->connected ? c->dev_copy : nullptr; }

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
  if (c->dev_copy == nullptr) WOQ_HIP(hipMalloc((void**)&c->dev_copy, sizeof(woq::CommDev)));
  WOQ_HIP(hipMemcpy(c->dev_copy, &c->dev, sizeof(woq::CommDev), hipMemcpyHostToDevice));
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
  if (c->dev_copy != nullptr)
    WOQ_HIP(hipMemcpy(c->dev_copy, &c->dev, sizeof(woq::CommDev), hipMemcpyHostToDevice));
  WOQ_END
}

void woq_comm_destroy(woq_comm* c) {
  if (!c) return;
  for (int r = 0; r < WOQ_COMM_MAX_WORLD; ++r)
    if (c->opened[r]) hipIpcCloseMemHandle(c->opened[r]);
  if (c->dev.peer[c->dev.rank]) hipFree(c->dev.peer[c->dev.rank]);
  if (c->dev.ctl) hipFree(c->dev.ctl);
  if (c->dev_copy) hipFree(c->dev_copy);
  delete c;
}

}  // extern "C"
