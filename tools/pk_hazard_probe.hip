// Development probe for the wrong-value defect of DESIGN.md section 6 (standalone: hipcc --offload-arch=gfx950 -O3 -o
// tools/pk_hazard_probe tools/pk_hazard_probe.hip; no torch, no liblmx).
//
// mask_post's plain-load build computes its bilinear taps with SLP-packed f32 VALU, and hipcc emits
//     v_pk_mul_f32 r, w, a ; s_nop 0 ; v_pk_add_f32 r, t, r op_sel:[0,1] op_sel_hi:[1,0] ; v_pk_mul_f32 r, wy, r ; s_nop 0 ;
//     v_add_f32 v, r.hi, r.lo
// (the high result of a packed op consumed by the first pass of the next one, one wait state apart), while the builds
// with atomic loads - the ones never seen wrong - happen to get the op_sel on the producer and no such consumer.  This
// probe runs exactly that instruction sequence (inline asm, NOPS wait states) in waves 4-7 of a 512-thread workgroup
// against a scalar v_mul/v_add evaluation of the same expression, bit for bit, while waves 0-3 (the SIMD partners)
// run what the aggressor kernel runs: a dense stream of 16x16x32 f16 MFMAs (MODE 1: two independent chains per
// operand read, like attn_kernel<QB=2>; MODE 2: one chain; MODE 0: partners idle).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lcg(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}
__device__ __forceinline__ float rnd(unsigned& s) { return (float)((int)(lcg(s) >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }

// The final v_add_f32 needs the two halves of a 64-bit pair as separate operands; inline asm cannot name sub-registers of
// an operand, so the chain is written over explicit scalar registers instead.
template <int NOPS>
__device__ __forceinline__ float pk_chain_regs(float w0, float w1, float a0, float a1, float b0, float b1, float y0, float y1) {
  float v;
  // v[40:41] = w, v[42:43] = a, v[44:45] = b, v[46:47] = wy, v[48:49] = t, v[50:51] = r
  if (NOPS == 0) {
    asm volatile(
        "v_mov_b32 v40, %1\n v_mov_b32 v41, %2\n v_mov_b32 v42, %3\n v_mov_b32 v43, %4\n"
        "v_mov_b32 v44, %5\n v_mov_b32 v45, %6\n v_mov_b32 v46, %7\n v_mov_b32 v47, %8\n"
        "s_nop 4\n"
        "v_pk_mul_f32 v[48:49], v[40:41], v[42:43]\n"
        "v_pk_mul_f32 v[50:51], v[40:41], v[44:45]\n"
        "s_nop 0\n"
        "v_pk_add_f32 v[50:51], v[48:49], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]\n"
        "v_mov_b32 v48, 0\n"
        "v_pk_mul_f32 v[50:51], v[46:47], v[50:51]\n"
        "s_nop 0\n"
        "v_add_f32 %0, v51, v50\n"
        : "=v"(v)
        : "v"(w0), "v"(w1), "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(y0), "v"(y1)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
  } else if (NOPS == -1) {  // no wait states at all (what the hardware would do without the compiler's hazard pass)
    asm volatile(
        "v_mov_b32 v40, %1\n v_mov_b32 v41, %2\n v_mov_b32 v42, %3\n v_mov_b32 v43, %4\n"
        "v_mov_b32 v44, %5\n v_mov_b32 v45, %6\n v_mov_b32 v46, %7\n v_mov_b32 v47, %8\n"
        "s_nop 4\n"
        "v_pk_mul_f32 v[48:49], v[40:41], v[42:43]\n"
        "v_pk_mul_f32 v[50:51], v[40:41], v[44:45]\n"
        "v_pk_add_f32 v[50:51], v[48:49], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]\n"
        "v_pk_mul_f32 v[50:51], v[46:47], v[50:51]\n"
        "v_add_f32 %0, v51, v50\n"
        : "=v"(v)
        : "v"(w0), "v"(w1), "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(y0), "v"(y1)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
  } else {  // generous wait states
    asm volatile(
        "v_mov_b32 v40, %1\n v_mov_b32 v41, %2\n v_mov_b32 v42, %3\n v_mov_b32 v43, %4\n"
        "v_mov_b32 v44, %5\n v_mov_b32 v45, %6\n v_mov_b32 v46, %7\n v_mov_b32 v47, %8\n"
        "s_nop 4\n"
        "v_pk_mul_f32 v[48:49], v[40:41], v[42:43]\n"
        "v_pk_mul_f32 v[50:51], v[40:41], v[44:45]\n"
        "s_nop 7\n"
        "v_pk_add_f32 v[50:51], v[48:49], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]\n"
        "s_nop 7\n"
        "v_pk_mul_f32 v[50:51], v[46:47], v[50:51]\n"
        "s_nop 7\n"
        "v_add_f32 %0, v51, v50\n"
        : "=v"(v)
        : "v"(w0), "v"(w1), "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(y0), "v"(y1)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
  }
  return v;
}

__device__ __forceinline__ float scalar_chain(float w0, float w1, float a0, float a1, float b0, float b1, float y0, float y1) {
  // t = (w0*a0, w1*a1); r = (w0*b0, w1*b1); r' = (t0 + r1, t1 + r0); r'' = (y0*r'0, y1*r'1); v = r''1 + r''0
  float t0, t1, r0, r1, s0, s1, v;
  asm volatile(
      "v_mul_f32 %0, %7, %9\n v_mul_f32 %1, %8, %10\n v_mul_f32 %2, %7, %11\n v_mul_f32 %3, %8, %12\n"
      "s_nop 4\n"
      "v_add_f32 %4, %0, %3\n v_add_f32 %5, %1, %2\n"
      "s_nop 4\n"
      "v_mul_f32 %4, %13, %4\n v_mul_f32 %5, %14, %5\n"
      "s_nop 4\n"
      "v_add_f32 %6, %5, %4\n"
      : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1), "=&v"(s0), "=&v"(s1), "=&v"(v)
      : "v"(w0), "v"(w1), "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(y0), "v"(y1));
  return v;
}

template <int NOPS, int MODE>
__global__ __launch_bounds__(512) void probe_kernel(unsigned long long* out, float* sink, int iters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (MODE == 0) return;
    half8 a, b;
    unsigned s = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
    for (int i = 0; i < 8; ++i) {
      a[i] = (_Float16)rnd(s);
      b[i] = (_Float16)rnd(s);
    }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    for (int it = 0; it < iters * 4; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      if (MODE == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
    }
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
    return;
  }
  unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  unsigned long long bad = 0, first = 0;
  for (int it = 0; it < iters; ++it) {
    const float lx = rnd(s) * 0.25f + 0.5f, ly = rnd(s) * 0.25f + 0.5f;
    const float a0 = rnd(s), a1 = rnd(s), b0 = rnd(s), b1 = rnd(s);
    const float got = pk_chain_regs<NOPS>(1.f - lx, lx, a0, a1, b0, b1, 1.f - ly, ly);
    const float want = scalar_chain(1.f - lx, lx, a0, a1, b0, b1, 1.f - ly, ly);
    if (__float_as_uint(got) != __float_as_uint(want)) {
      if (!bad) first = ((unsigned long long)__float_as_uint(got) << 32) | __float_as_uint(want);
      ++bad;
    }
  }
  if (bad) {
    atomicAdd(&out[0], bad);
    atomicAdd(&out[1], 1ull);
    out[2] = first;
    out[3] = ((unsigned long long)blockIdx.x << 32) | (unsigned)(wave * 64 + lane);
  }
}

template <int NOPS, int MODE>
static void run(const char* label, int blocks, int iters) {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, (size_t)blocks * 256 * 4);
  hipMemset(out, 0, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe_kernel<NOPS, MODE>), dim3(blocks), dim3(512), 0, 0, out, sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4];
  hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  printf("%-44s blocks %d iters %d: %llu wrong results in %llu lanes (first got/want %016llx at %016llx) of %.3g evaluations, %.2f ms\n", label,
         blocks, iters, h[0], h[1], h[2], h[3], (double)blocks * 256 * iters, ms);
  fflush(stdout);
  hipFree(out);
  hipFree(sink);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 20000;
  run<0, 0>("compiler wait states, partners idle", blocks, iters);
  run<0, 2>("compiler wait states, partners 1 MFMA chain", blocks, iters);
  run<0, 1>("compiler wait states, partners 2 MFMA chains", blocks, iters);
  run<-1, 0>("no wait states, partners idle", blocks, iters);
  run<-1, 1>("no wait states, partners 2 MFMA chains", blocks, iters);
  run<7, 1>("s_nop 7 everywhere, partners 2 MFMA chains", blocks, iters);
  return 0;
}
