// Shared host-side helpers of libmillieye_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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

// Store policy of the conv epilogues: 0 = plain (default), 1 = nt, 2 = sc1 for 16-byte stores (4-byte stores use nt in both
// modes: a scalar sc1 store is one fabric write each); MILLIEYE_STORE_MODE selects (A/B measurements).  Round 3 measured
// write-through stores because a layer's output waits dirty in the XCD's L2 for the end-of-kernel release: on the 52 x 52
// 3x3 layer launched back to back the idle time between the last workgroup of one launch and the first of the next drops
// from 3.4 to 2.1 us - but inside the network the next layer READS that output, and dropping it from the L2 costs more than
// the shorter boundary buys (bf16 pipeline @32: plain 7970-7990, sc1 7835, nt 6820 frames/s; fp32 unchanged) - so plain
// stays (profiles/r03_kernel_evolution.md).
inline int store_mode() {
  static const int mode = [] {
    const char* e = getenv("MILLIEYE_STORE_MODE");
    return e ? atoi(e) : 0;
  }();
  return mode;
}

typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* ptr, uint4 v, int mode) {
  if (mode == 0) {
    *reinterpret_cast<uint4*>(ptr) = v;
  } else {
    st_u32x4 d = {v.x, v.y, v.z, v.w};
    if (mode == 1)
      asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(d) : "memory");
    else
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(d) : "memory");
  }
}
__device__ __forceinline__ void store4(float* ptr, float v, int mode) {
  if (mode == 0)
    *ptr = v;
  else
    __builtin_nontemporal_store(v, ptr);
}

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
