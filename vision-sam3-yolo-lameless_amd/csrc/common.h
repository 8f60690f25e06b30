// common.h — shared helpers for the liblmx HIP sources (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/lmx.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// thread-local error text (lmx_last_error)
void lmx_set_error(const char* fmt, ...);

#define LMX_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      lmx_set_error(__VA_ARGS__);         \
      return LMX_EINVAL;                  \
    }                                     \
  } while (0)

#define LMX_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      lmx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return LMX_EHIP;                                                                  \
    }                                                                                   \
  } while (0)

static inline int lmx_launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    lmx_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return LMX_EHIP;
  }
  return LMX_OK;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
