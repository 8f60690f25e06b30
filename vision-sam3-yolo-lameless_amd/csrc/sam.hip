// sam.hip — SAM / Hiera pieces that are not GEMM- or attention-shaped (SURVEY.md K9, K10 im2col, K15 Q-pool).
// HBM-bound element work, one pass, 16-byte stores.
#include "common.h"

namespace {

inline int grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// thread = one 8-column chunk of one output row (token); builds the 8 normalised values and stores 16 bytes.
__global__ __launch_bounds__(256) void im2col_u8_kernel(const uint8_t* __restrict__ img, const float* __restrict__ lut,
                                                        half_t* __restrict__ out, int n, int rh, int rw, int OH, int OW, int KH,
                                                        int KW, int stride, int pad, int64_t ldo) {
  const int chunks = (int)(ldo / 8);
  const int K = KH * KW * 3;
  const int64_t total = (int64_t)n * OH * OW * chunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % chunks);
    const int64_t tok = i / chunks;
    const int ox = (int)(tok % OW);
    const int64_t r = tok / OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    half8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ck * 8 + e;
      float val = 0.f;
      if (k < K) {
        const int tap = k / 3, c = k - tap * 3;
        const int ky = tap / KW, kx = tap - ky * KW;
        const int y = oy * stride - pad + ky, x = ox * stride - pad + kx;
        if ((unsigned)y < (unsigned)rh && (unsigned)x < (unsigned)rw)
          val = lut[c * 256 + img[(((int64_t)b * rh + y) * rw + x) * 3 + c]];
      }
      v[e] = (half_t)val;
    }
    *reinterpret_cast<half8_t*>(out + tok * ldo + ck * 8) = v;
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool2_kernel(const T* __restrict__ src, int64_t lds, T* __restrict__ dst, int64_t ldd,
                                                       int n, int H, int W, int C) {
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const int cc = C / VEC;
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)n * Ho * Wo * cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc) * VEC;
    const int64_t p = i / cc;
    const int x = (int)(p % Wo);
    const int64_t r = p / Wo;
    const int y = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const T* s = src + (((int64_t)b * H + 2 * y) * W + 2 * x) * lds + c;
    const vec_t a0 = *reinterpret_cast<const vec_t*>(s), a1 = *reinterpret_cast<const vec_t*>(s + lds);
    const vec_t a2 = *reinterpret_cast<const vec_t*>(s + (int64_t)W * lds), a3 = *reinterpret_cast<const vec_t*>(s + (int64_t)W * lds + lds);
    vec_t m;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      T t0 = a0[e] > a1[e] ? a0[e] : a1[e];
      T t1 = a2[e] > a3[e] ? a2[e] : a3[e];
      m[e] = t0 > t1 ? t0 : t1;
    }
    *reinterpret_cast<vec_t*>(dst + p * ldd + c) = m;
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, int64_t lds, half_t* __restrict__ dst, int64_t ldd,
                                                   int64_t rows, int cols) {
  const int cc = cols / 4;
  const int64_t total = rows * cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc) * 4;
    const int64_t r = i / cc;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds + c);
    half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *reinterpret_cast<half4_t*>(dst + r * ldd + c) = h;
  }
}

}  // namespace

extern "C" int lmx_k_im2col_u8(const uint8_t* img, const float* lut, void* out, int n, int rh, int rw, int IH, int IW, int KH,
                               int KW, int stride, int pad, int64_t ldo, lmx_stream_t stream) {
  LMX_REQUIRE(img && lut && out, "lmx_k_im2col_u8: null pointer");
  LMX_REQUIRE(n > 0 && rh > 0 && rw > 0 && rh <= IH && rw <= IW && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
              "lmx_k_im2col_u8: geometry");
  LMX_REQUIRE(ldo % 8 == 0 && ldo >= (int64_t)KH * KW * 3 && aligned16(out), "lmx_k_im2col_u8: ldo/alignment");
  const int OH = (IH + 2 * pad - KH) / stride + 1, OW = (IW + 2 * pad - KW) / stride + 1;
  hipLaunchKernelGGL(im2col_u8_kernel, dim3(grid_for((int64_t)n * OH * OW * (ldo / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), img, lut, reinterpret_cast<half_t*>(out), n, rh, rw, OH, OW, KH, KW,
                     stride, pad, ldo);
  return lmx_launch_check("im2col_u8_kernel");
}

extern "C" int lmx_k_maxpool2(const void* src, int64_t lds, void* dst, int64_t ldd, int dtype, int n, int H, int W, int C,
                              lmx_stream_t stream) {
  LMX_REQUIRE(src && dst, "lmx_k_maxpool2: null pointer");
  LMX_REQUIRE(n > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && lds >= C && ldd >= C, "lmx_k_maxpool2: shape");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == LMX_F16) {
    LMX_REQUIRE(C % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && aligned16(src) && aligned16(dst), "lmx_k_maxpool2: f16 alignment");
    hipLaunchKernelGGL((maxpool2_kernel<half_t, 8>), dim3(grid_for((int64_t)n * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, st,
                       reinterpret_cast<const half_t*>(src), lds, reinterpret_cast<half_t*>(dst), ldd, n, H, W, C);
  } else if (dtype == LMX_F32) {
    LMX_REQUIRE(C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst), "lmx_k_maxpool2: f32 alignment");
    hipLaunchKernelGGL((maxpool2_kernel<float, 4>), dim3(grid_for((int64_t)n * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, st,
                       reinterpret_cast<const float*>(src), lds, reinterpret_cast<float*>(dst), ldd, n, H, W, C);
  } else {
    LMX_REQUIRE(false, "lmx_k_maxpool2: dtype %d", dtype);
  }
  return lmx_launch_check("maxpool2_kernel");
}

extern "C" int lmx_k_cast_f32_f16(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int cols,
                                  lmx_stream_t stream) {
  LMX_REQUIRE(src && dst, "lmx_k_cast_f32_f16: null pointer");
  LMX_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && lds >= cols && ldd >= cols,
              "lmx_k_cast_f32_f16: shape");
  LMX_REQUIRE(aligned16(src) && ((((uintptr_t)dst) & 7) == 0), "lmx_k_cast_f32_f16: alignment");
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, lds,
                     reinterpret_cast<half_t*>(dst), ldd, rows, cols);
  return lmx_launch_check("cast_kernel");
}
