// woq_comm_dev.h — device side of the tensor-parallel exchange shared by the collective kernels (woq_comm.hip) and
// the kernels that push their partial sums themselves (woq_gemv_xqs.h epilogue): inbox layout, granule push / pull.
// Protocol and reasons: woq_comm.hip.
#pragma once
#include "woq_device.h"

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

}  // namespace woq
