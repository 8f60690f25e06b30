// exact.hip — the element-wise pieces of the EXACT-precision YOLO plan (lmx/yolo.py precision="exact").
//
// north_star asks for box indices and NMS keep-sets that are bit-exact against the fp32 CPU path of
// services/yolo-pipeline/app/main.py:76.  An f16 network deviates 3e-3 .. 6e-3 in score from it; fp32 on the VALU or the f32
// MFMA (157 TFLOP/s peak) would cost ~10x the f16 network.  The exact plan keeps the f16 MFMA kernels and gives them operands
// that carry 22 mantissa bits: a value x travels as the channel TRIPLE
//        [ hi | lo | hi ],   hi = f16(x),   lo = f16((x - hi) * 2048)            ("x3" format, per channel group of width g)
// and a weight row as [ whi | whi / 2048 | wlo ] (split once at load, rows pre-scaled by a power of two so that wlo stays a
// normal f16): ONE launch of the existing GEMM / 3x3 implicit-GEMM kernel over K' = 3K then accumulates
//        hi*whi + lo*(whi/2048) + hi*wlo  =  x*w  up to the dropped (x - hi)(w - whi) term, 2^-22 relative,
// in the f32 MFMA accumulators and writes the f32 pre-activation.  The kernels here are the glue around that launch:
//   split3_kernel       f32 pre-activation -> activation (+ x3 residual) -> x3 triple, written into a channel slice
//   maxpool5_x3_kernel  SPPF's max_pool2d(5,1,2) on x3 slices (maximum by VALUE, the pair travels with it)
//   stem_conv_x3        the f32 VALU stem writing x3
// nearest upsampling of an x3 slice is a plain copy: lmx_k_upsample2 with 3x the channels.
// All of it is HBM-bound element work: 16-byte lane accesses along the NHWC channel axis.
#include "common.h"

namespace {

inline int grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// x = hi + lo / 2048 with hi the nearest f16: (x - hi) is exact in f32 (it needs at most 13 of the 24 mantissa bits), the
// scaling by 2^11 keeps lo a NORMAL f16 down to |x| = 2^-13 (an unscaled lo is subnormal below 0.25)
__device__ __forceinline__ void split_hi_lo(float v, half_t& hi, half_t& lo) {
  hi = (half_t)v;
  lo = (half_t)((v - (float)hi) * 2048.0f);
}
__device__ __forceinline__ float join_hi_lo(half_t hi, half_t lo) { return (float)hi + (float)lo * (1.0f / 2048.0f); }

// SiLU as torch's CPU kernel writes it: x / (1 + exp(-x)) with a true division (the f16 plan's v_exp / v_rcp form is 1-2 ulp
// looser, which the exact plan cannot afford to spend)
__device__ __forceinline__ float act_exact(float v, int act) {
  if (act == LMX_ACT_SILU) return v / (1.0f + expf(-v));
  if (act == LMX_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == LMX_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // the erf form torch's GELU evaluates
  return v;
}

// rows x N f32 (row stride ldx) -> x3 groups of width g in a channel slice with pixel stride ldo (in f16 elements); an
// optional x3 residual (same grouping, pixel stride ldr) is added after the activation (C2f's shortcut: y = x + cv2(cv1(x)))
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int64_t ldx, int act, const half_t* __restrict__ res,
                                                     int64_t ldr, half_t* __restrict__ out, int64_t ldo, int64_t rows, int N, int g,
                                                     int nsum, int64_t sstride) {
  const int nc = N / 8;
  const int64_t total = rows * nc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n0 = (int)(i % nc) * 8;
    const int64_t m = i / nc;
    const int q = n0 / g, r = n0 - q * g;
    const int base = q * 3 * g + r;
    f32x4 a = *reinterpret_cast<const f32x4*>(x + m * ldx + n0);
    f32x4 b = *reinterpret_cast<const f32x4*>(x + m * ldx + n0 + 4);
    for (int sidx = 1; sidx < nsum; ++sidx) {  // split-K partials, added in index order
      a += *reinterpret_cast<const f32x4*>(x + sidx * sstride + m * ldx + n0);
      b += *reinterpret_cast<const f32x4*>(x + sidx * sstride + m * ldx + n0 + 4);
    }
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_exact(v[e], act);
    if (res) {
      const half8_t rh = *reinterpret_cast<const half8_t*>(res + m * ldr + base);
      const half8_t rl = *reinterpret_cast<const half8_t*>(res + m * ldr + base + g);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] + join_hi_lo(rh[e], rl[e]);
    }
    half8_t hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      half_t h, l;
      split_hi_lo(v[e], h, l);
      hi[e] = h;
      lo[e] = l;
    }
    half_t* o = out + m * ldo + base;
    *reinterpret_cast<half8_t*>(o) = hi;
    *reinterpret_cast<half8_t*>(o + g) = lo;
    *reinterpret_cast<half8_t*>(o + 2 * g) = hi;
  }
}

// max_pool2d(5, 1, 2) on an x3 slice of C logical channels (3C f16 channels, pixel strides lds / ldd)
__global__ __launch_bounds__(256) void maxpool5_x3_kernel(const half_t* __restrict__ src, int64_t lds, half_t* __restrict__ dst,
                                                          int64_t ldd, int n, int H, int W, int C) {
  const int cc = C / 8;
  const int64_t total = (int64_t)n * H * W * cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc) * 8;
    const int64_t p = i / cc;
    const int x = (int)(p % W);
    const int64_t r = p / W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float mv[8];
    half8_t mh, ml;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mv[e] = -INFINITY;
      mh[e] = (half_t)(-65504.0f);
      ml[e] = (half_t)0.0f;
    }
    for (int dy = -2; dy <= 2; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = -2; dx <= 2; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const half_t* s = src + (((int64_t)b * H + yy) * W + xx) * lds + c;
        const half8_t h = *reinterpret_cast<const half8_t*>(s);
        const half8_t l = *reinterpret_cast<const half8_t*>(s + C);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = join_hi_lo(h[e], l[e]);  // exact: 22 bits fit an f32
          if (v > mv[e]) {
            mv[e] = v;
            mh[e] = h[e];
            ml[e] = l[e];
          }
        }
      }
    }
    half_t* o = dst + p * ldd + c;
    *reinterpret_cast<half8_t*>(o) = mh;
    *reinterpret_cast<half8_t*>(o + C) = ml;
    *reinterpret_cast<half8_t*>(o + 2 * C) = mh;
  }
}

// the stem of yolo.hip (Conv(3 -> Cout, k3, s2, p1) + bias + SiLU on the u8 letterboxed frame, f32 FMA) writing x3
template <int CO_T>
__global__ __launch_bounds__(256) void stem_conv_x3_kernel(const uint8_t* __restrict__ img, const float* __restrict__ w,
                                                           const float* __restrict__ bias, half_t* __restrict__ out, int n, int H,
                                                           int W, int Ho, int Wo, int Cout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ws = reinterpret_cast<float*>(smem);  // [27][Cout] + bias[Cout]
  for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) ws[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) ws[27 * Cout + i] = bias[i];
  __syncthreads();
  const int64_t total = (int64_t)n * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const int64_t r = i / Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const uint8_t* s = img + (((int64_t)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) x[(ky * 3 + kx) * 3 + c] = ok ? (float)s[c] / 255.0f : 0.0f;
      }
    half_t* o = out + i * 3 * Cout;
    for (int c0 = 0; c0 < Cout; c0 += CO_T) {
      float acc[CO_T];
#pragma unroll
      for (int j = 0; j < CO_T; ++j) acc[j] = 0.0f;
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const float xv = x[t];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(xv, ws[t * Cout + c0 + j], acc[j]);
      }
      half8_t hv, lv;
#pragma unroll
      for (int j = 0; j < CO_T; ++j) {
        const float z = acc[j] + ws[27 * Cout + c0 + j];  // bias last, as Conv2d adds it to the finished sum
        half_t h, l;
        split_hi_lo(z / (1.0f + expf(-z)), h, l);
        hv[j] = h;
        lv[j] = l;
      }
      *reinterpret_cast<half8_t*>(o + c0) = hv;
      *reinterpret_cast<half8_t*>(o + Cout + c0) = lv;
      *reinterpret_cast<half8_t*>(o + 2 * Cout + c0) = hv;
    }
  }
}

// ---- the exact plan of the SAM mask decoder (lmx/sam_decoder.py precision="exact"): its attentions are tiny (7 tokens against 4096
// image positions, head dims 16 / 32) and run in plain f32 on the VALU; TF:models/sam/modeling_sam.py:205-268 SamAttention.
// Tk > 16: one wave per (batch, head, query): lanes stride over the keys, two passes (maximum, then exp / sum / weighted V:
// the dot product is recomputed instead of stored), wave reductions at the end.
// Tk <= 16: one thread per (batch, head, query).
template <int HD>
__global__ __launch_bounds__(256) void attn_f32_wave_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                            int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                                            float* __restrict__ O, int64_t ldo, int B, int H, int Tq, int Tk, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)B * H * Tq) return;
  const int tq = (int)(item % Tq);
  const int64_t bh = item / Tq;
  const int h = (int)(bh % H), b = (int)(bh / H);
  float q[HD];
  const float* qp = Q + ((int64_t)b * Tq + tq) * ldq + h * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(qp + d);
    q[d] = v[0], q[d + 1] = v[1], q[d + 2] = v[2], q[d + 3] = v[3];
  }
  const float* Kb = K + (int64_t)b * Tk * ldk + h * HD;
  const float* Vb = V + (int64_t)b * Tk * ldv + h * HD;
  auto score = [&](int j) {
    const float* kp = Kb + (int64_t)j * ldk;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(kp + d);
      s += q[d] * v[0] + q[d + 1] * v[1] + q[d + 2] * v[2] + q[d + 3] * v[3];
    }
    return s * scale;
  };
  float mx = -INFINITY;
  for (int j = lane; j < Tk; j += 64) mx = fmaxf(mx, score(j));
  mx = wave_max(mx);
  float l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int j = lane; j < Tk; j += 64) {
    const float e = expf(score(j) - mx);
    l += e;
    const float* vp = Vb + (int64_t)j * ldv;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(vp + d);
      o[d] += e * v[0], o[d + 1] += e * v[1], o[d + 2] += e * v[2], o[d + 3] += e * v[3];
    }
  }
  l = wave_sum(l);
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = wave_sum(o[d]);
  if (lane == 0) {
    float* op = O + ((int64_t)b * Tq + tq) * ldo + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = o[d] / l;
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_thread_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ K,
                                                              int64_t ldk, const float* __restrict__ V, int64_t ldv,
                                                              float* __restrict__ O, int64_t ldo, int B, int H, int Tq, int Tk, float scale) {
  const int64_t total = (int64_t)B * H * Tq;
  for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(item % H);  // heads fastest: the 8 heads of a query read one contiguous row
    const int64_t bq = item / H;
    const int tq = (int)(bq % Tq), b = (int)(bq / Tq);
    float q[HD];
    const float* qp = Q + ((int64_t)b * Tq + tq) * ldq + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = qp[d];
    float s[16];
    float mx = -INFINITY;
    for (int j = 0; j < Tk; ++j) {
      const float* kp = K + ((int64_t)b * Tk + j) * ldk + h * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a += q[d] * kp[d];
      s[j] = a * scale;
      mx = fmaxf(mx, s[j]);
    }
    float l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int j = 0; j < Tk; ++j) {
      const float e = expf(s[j] - mx);
      l += e;
      const float* vp = V + ((int64_t)b * Tk + j) * ldv + h * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] += e * vp[d];
    }
    float* op = O + ((int64_t)b * Tq + tq) * ldo + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = o[d] / l;
  }
}

// lmx_k_hyper_mask on an f32 upscaled embedding, with the upscaler's last GELU applied on load (exact erf form)
__global__ __launch_bounds__(256) void hyper_mask_f32_kernel(const float* __restrict__ up, const float* __restrict__ hyper,
                                                             float* __restrict__ logits, int n, int G, int C, int act) {
  const int64_t total = (int64_t)n * G * G * 16;
  const int S = 4 * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q2 = (int)(i & 3), q1 = (int)((i >> 2) & 3);
    const int64_t cell = i >> 4;
    const int x = (int)(cell % G);
    const int64_t r = cell / G;
    const int y = (int)(r % G);
    const int b = (int)(r / G);
    const float* u = up + i * C;
    const float* hy = hyper + (int64_t)b * C;
    float acc = 0.f;
    for (int c = 0; c < C; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(u + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += act_exact(v[e], act) * hy[c + e];
    }
    const int Y = 4 * y + 2 * (q1 >> 1) + (q2 >> 1), X = 4 * x + 2 * (q1 & 1) + (q2 & 1);
    logits[((int64_t)b * S + Y) * S + X] = acc;
  }
}

}  // namespace

extern "C" int lmx_k_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o,
                                   int64_t ldo, int B, int H, int Tq, int Tk, int hd, float scale, lmx_stream_t stream) {
  LMX_REQUIRE(q && k && v && o, "lmx_k_attention_f32: null pointer");
  LMX_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0 && (hd == 16 || hd == 32), "lmx_k_attention_f32: head dim %d (16 or 32)", hd);
  LMX_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldq >= (int64_t)H * hd && ldk >= (int64_t)H * hd && ldv >= (int64_t)H * hd &&
                  ldo >= (int64_t)H * hd,
              "lmx_k_attention_f32: strides");
  LMX_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v), "lmx_k_attention_f32: alignment");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t items = (int64_t)B * H * Tq;
  LMX_REQUIRE(items < (1ll << 31), "lmx_k_attention_f32: too many queries");
  if (Tk <= 16) {
    if (hd == 16)
      hipLaunchKernelGGL((attn_f32_thread_kernel<16>), dim3(grid_for(items)), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, o, ldo, B, H, Tq, Tk, scale);
    else
      hipLaunchKernelGGL((attn_f32_thread_kernel<32>), dim3(grid_for(items)), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, o, ldo, B, H, Tq, Tk, scale);
    return lmx_launch_check("attn_f32_thread_kernel");
  }
  const unsigned grid = (unsigned)((items + 3) / 4);
  if (hd == 16)
    hipLaunchKernelGGL((attn_f32_wave_kernel<16>), dim3(grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, o, ldo, B, H, Tq, Tk, scale);
  else
    hipLaunchKernelGGL((attn_f32_wave_kernel<32>), dim3(grid), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, o, ldo, B, H, Tq, Tk, scale);
  return lmx_launch_check("attn_f32_wave_kernel");
}

extern "C" int lmx_k_hyper_mask_f32(const float* up, const float* hyper, float* logits, int n, int G, int C, int act,
                                    lmx_stream_t stream) {
  LMX_REQUIRE(up && hyper && logits, "lmx_k_hyper_mask_f32: null pointer");
  LMX_REQUIRE(n > 0 && G > 0 && C > 0 && C % 4 == 0 && C <= 64 && aligned16(up), "lmx_k_hyper_mask_f32: shape");
  LMX_REQUIRE(act == LMX_ACT_NONE || act == LMX_ACT_GELU, "lmx_k_hyper_mask_f32: activation %d", act);
  hipLaunchKernelGGL(hyper_mask_f32_kernel, dim3(grid_for((int64_t)n * G * G * 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     up, hyper, logits, n, G, C, act);
  return lmx_launch_check("hyper_mask_f32_kernel");
}

extern "C" int lmx_k_split3(const float* x, int64_t ldx, int act, const void* res3, int64_t ldr, void* out3, int64_t ldo,
                            int64_t rows, int N, int g, int nsum, int64_t sum_stride, lmx_stream_t stream) {
  LMX_REQUIRE(x && out3, "lmx_k_split3: null pointer");
  LMX_REQUIRE(rows > 0 && N > 0 && g > 0 && g % 8 == 0 && N % g == 0, "lmx_k_split3: N=%d must be whole groups of g=%d (g %% 8 == 0)", N, g);
  LMX_REQUIRE(ldx % 4 == 0 && ldx >= N && ldo % 8 == 0 && ldo >= 3 * (int64_t)N, "lmx_k_split3: strides (ldx %lld, ldo %lld)",
              (long long)ldx, (long long)ldo);
  LMX_REQUIRE(aligned16(x) && aligned16(out3), "lmx_k_split3: alignment");
  LMX_REQUIRE(act >= LMX_ACT_NONE && act <= LMX_ACT_RELU, "lmx_k_split3: activation %d", act);
  if (res3) LMX_REQUIRE(ldr % 8 == 0 && ldr >= 3 * (int64_t)N && aligned16(res3), "lmx_k_split3: residual stride / alignment");
  LMX_REQUIRE(nsum >= 1 && nsum <= 64 && (nsum == 1 || sum_stride % 4 == 0), "lmx_k_split3: nsum=%d sum_stride=%lld", nsum, (long long)sum_stride);
  hipLaunchKernelGGL(split3_kernel, dim3(grid_for(rows * (N / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, act,
                     reinterpret_cast<const half_t*>(res3), ldr, reinterpret_cast<half_t*>(out3), ldo, rows, N, g, nsum, sum_stride);
  return lmx_launch_check("split3_kernel");
}

extern "C" int lmx_k_maxpool5_x3(const void* src3, int64_t lds, void* dst3, int64_t ldd, int n, int H, int W, int C,
                                 lmx_stream_t stream) {
  LMX_REQUIRE(src3 && dst3, "lmx_k_maxpool5_x3: null pointer");
  LMX_REQUIRE(n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && lds >= 3 * (int64_t)C &&
                  ldd >= 3 * (int64_t)C,
              "lmx_k_maxpool5_x3: shape");
  LMX_REQUIRE(aligned16(src3) && aligned16(dst3), "lmx_k_maxpool5_x3: alignment");
  hipLaunchKernelGGL(maxpool5_x3_kernel, dim3(grid_for((int64_t)n * H * W * (C / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const half_t*>(src3), lds,
                     reinterpret_cast<half_t*>(dst3), ldd, n, H, W, C);
  return lmx_launch_check("maxpool5_x3_kernel");
}

extern "C" int lmx_k_stem_conv_x3(const uint8_t* img, const float* w, const float* bias, void* out3, int n, int H, int W, int Cout,
                                  lmx_stream_t stream) {
  LMX_REQUIRE(img && w && bias && out3, "lmx_k_stem_conv_x3: null pointer");
  LMX_REQUIRE(n > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && Cout <= 256, "lmx_k_stem_conv_x3: shape (Cout %d)", Cout);
  LMX_REQUIRE(aligned16(out3), "lmx_k_stem_conv_x3: out alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t smem = (size_t)28 * Cout * sizeof(float);
  hipLaunchKernelGGL((stem_conv_x3_kernel<8>), dim3(grid_for((int64_t)n * Ho * Wo)), dim3(256), smem,
                     reinterpret_cast<hipStream_t>(stream), img, w, bias, reinterpret_cast<half_t*>(out3), n, H, W, Ho, Wo, Cout);
  return lmx_launch_check("stem_conv_x3_kernel");
}
