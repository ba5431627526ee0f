// Shared host-side helpers of libmillieye_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "millieye_hip.h"

namespace me {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace me

#define ME_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      me::set_error(__VA_ARGS__);     \
      return (code);                  \
    }                                 \
  } while (0)

#define ME_HIP(call)                                                 \
  do {                                                               \
    hipError_t e__ = (call);                                         \
    if (e__ != hipSuccess) {                                         \
      me::set_error("%s failed: %s", #call, hipGetErrorString(e__)); \
      return (int)e__;                                               \
    }                                                                \
  } while (0)
