// woq_launch.h — host-side error plumbing for the C ABI (thread-local message, int status).
// The reference reports errors as TORCH_CHECK -> c10::Error -> Python RuntimeError with a "QBits:"
// prefix (bestla_weightonly_dispatcher.cpp:289,368); this keeps the same text convention across a
// plain-C boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/woq_hip.h"

namespace woq {
std::string& last_error_ref();
inline int fail(const std::string& msg) {
  last_error_ref() = msg;
  return 1;
}
}  // namespace woq

namespace woq {
// Scratch for one call: from the caller's workspace (woq_set_workspace — the reference's set_woq_workspace contract,
// qbits.cpp:142-144, bestla_weightonly_dispatcher.cpp:394-397: a raw pointer into a caller-owned tensor that outlives
// every call) when it has room, which keeps the call allocation-free and capturable into a hipGraph; otherwise
// stream-ordered allocation, like the reference's per-call amalloc (:108-118,179). Nested takes release in LIFO order.
// Like the reference's global workspace pointer, not safe for concurrent calls from several host threads / streams.
void* scratch_take(size_t bytes, hipStream_t st, bool* own);
void scratch_release(void* p, size_t bytes, bool own, hipStream_t st);
}  // namespace woq

#define WOQ_TRY try {
#define WOQ_END                                                \
  return 0;                                                    \
  }                                                            \
  catch (const std::exception& ex) {                           \
    return woq::fail(std::string("QBits: ") + ex.what());     \
  }
#define WOQ_FAIL(msg) return woq::fail(msg)
#define WOQ_CHECK(cond, msg) \
  do {                       \
    if (!(cond)) return woq::fail(msg); \
  } while (0)
#define WOQ_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return woq::fail(std::string("QBits: HIP error '") + hipGetErrorString(_e) + "' at " #expr); \
  } while (0)
