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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// erf-GELU for two values at once: gelu(v) = v * (0.5 + g(v)), g(v) = 0.5*erf(v/sqrt2) ~ vc * P(vc^2) with
// vc = clamp(v, -4.5, 4.5) and P of degree 9 (weighted least-squares Chebyshev fit, evaluated in f32: |error| <= 3.4e-5
// for every v, i.e. a tenth of the f16 rounding of an O(1) result; tools/fit_gelu.py reproduces the coefficients).
// No transcendental (v_rcp / v_exp are quarter rate) and written on 2-vectors so the Horner chain is v_pk_fma_f32:
// ~7.5 VALU issue slots per value against ~23 for the Abramowitz-Stegun erf it replaces.  (The same polynomial on scalar v_fma_f32
// chains — identical bits — measured the same: fused LN + MLP 0.92 / 0.63 ms against 0.88 / 0.64, bench 596.8 against 597.3
// frames/s, profiles/r03_gelu_scalar_ab.txt.)  At D = 112..448 the GELU of an
// MLP otherwise costs more VALU time than its two GEMMs cost MFMA time.
__device__ __forceinline__ f32x2 gelu_pk(f32x2 v) {
  f32x2 vc;
  vc[0] = __builtin_amdgcn_fmed3f(v[0], -4.5f, 4.5f);
  vc[1] = __builtin_amdgcn_fmed3f(v[1], -4.5f, 4.5f);
  const f32x2 u = vc * vc;
  f32x2 p = {-1.223331157e-12f, -1.223331157e-12f};
  p = p * u + 1.524351007e-10f;
  p = p * u + -8.481037793e-09f;
  p = p * u + 2.798429583e-07f;
  p = p * u + -6.151207192e-06f;
  p = p * u + 9.615961466e-05f;
  p = p * u + -1.114801972e-03f;
  p = p * u + 9.814679350e-03f;
  p = p * u + -6.632144717e-02f;
  p = p * u + 3.988863025e-01f;
  const f32x2 g = vc * p;
  return v * g + 0.5f * v;
}
__device__ __forceinline__ float gelu_1(float v) {
  const f32x2 r = gelu_pk(f32x2{v, v});
  return r[0];
}

// The activation of every GEMM / conv epilogue.  ONE definition for all kernel variants: which variant the launcher picks
// depends on M = frames x pixels, so two variants that rounded differently would make a frame's result depend on the batch
// it rides in (tools/batch_invariance_probe.py).  SiLU by v_exp / v_rcp, erf-GELU by the polynomial above.
__device__ __forceinline__ float lmx_act(float v, int act) {
  if (act == LMX_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.44269504088896340736f));
  if (act == LMX_ACT_GELU) return gelu_1(v);
  if (act == LMX_ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One LDS-DMA wave-instruction (buffer_load_dwordx4 ... lds): lane i's 16 bytes land at dst + 16*i.  Kept in a
// NON-template function: inside a kernel template with dependent arguments the amdgcn builtin makes the HOST pass drop
// the kernel's instantiation without a diagnostic (the .so then fails to load with an undefined __device_stub__).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* dst_lds, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst_lds, 16, voffset, soffset, 0, 0);
}

// counted wait with a literal immediate per instantiation (an "n"-constrained template-dependent asm operand makes
// hipcc drop the HOST stub of the enclosing kernel template without a diagnostic)
#define LMX_WAIT_CASE(n) \
  if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N == 0 || N == 1 || N == 2 || N == 3 || N == 4 || N == 6 || N == 8 || N == 9 || N == 12 || N == 16 || N == 18 || N == 24 ||
                    N == 32,
                "add the literal for this count");
  LMX_WAIT_CASE(0);
  LMX_WAIT_CASE(1);
  LMX_WAIT_CASE(2);
  LMX_WAIT_CASE(3);
  LMX_WAIT_CASE(4);
  LMX_WAIT_CASE(6);
  LMX_WAIT_CASE(8);
  LMX_WAIT_CASE(9);
  LMX_WAIT_CASE(12);
  LMX_WAIT_CASE(16);
  LMX_WAIT_CASE(18);
  LMX_WAIT_CASE(24);
  LMX_WAIT_CASE(32);
}
#undef LMX_WAIT_CASE
// wait until at most y * PT of this wave's vector-memory operations are outstanding (y = 0..4 whole k-tiles of PT DMAs)
template <int PT>
__device__ __forceinline__ void wait_tiles(int y) {
  if (y >= 4)
    wait_vmcnt<4 * PT>();
  else if (y == 3)
    wait_vmcnt<3 * PT>();
  else if (y == 2)
    wait_vmcnt<2 * PT>();
  else if (y == 1)
    wait_vmcnt<PT>();
  else
    wait_vmcnt<0>();
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
