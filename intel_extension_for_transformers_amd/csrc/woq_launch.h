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
