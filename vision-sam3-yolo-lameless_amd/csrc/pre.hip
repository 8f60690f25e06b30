// pre.hip — frame preprocessing (K9/K21): Pillow-exact separable u8 resampling and the crop/normalise/patchify
// pass that writes the ViT patch matrix.  HBM-bound byte work: a 1080p BGR frame is 6.2 MB in, the outputs are
// small, so the figure of merit is input bytes / time.
//
// Pillow reference (not in this tree; Pillow 10+ src/libImaging/Resample.c): ImagingResampleHorizontal_8bpc /
// ImagingResampleVertical_8bpc with PRECISION_BITS = 22: acc starts at 1<<21, adds u8*coef (int32 wraps never
// reached: |coef| < 2^22*1.x, <= ~30 taps), result = clip8(acc >> 22).
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= PRECISION_BITS;  // arithmetic shift, like the C code's signed >>
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// workgroup = one source row: the row (sw*3 bytes, 5760 for 1080p) is staged into LDS with 16-byte coalesced loads, then
// every thread produces output pixels of that row from LDS.  (One thread per output pixel reading its taps straight from
// global memory issued 3 byte loads per tap on overlapping 12..24-byte windows: 0.4 TB/s.)
constexpr int RESIZE_H_MAX_ROW = 16384;  // bytes of LDS for one source row (sw <= 5461)
__global__ __launch_bounds__(256) void pil_resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                           int n, int sh, int sw, int dw,
                                                           const int32_t* __restrict__ bounds,
                                                           const int32_t* __restrict__ kk, int ksize, int swap_rb) {
  __shared__ __attribute__((aligned(16))) uint8_t rowbuf[RESIZE_H_MAX_ROW];
  const int rowb = sw * 3;
  for (int64_t row = blockIdx.x; row < (int64_t)n * sh; row += gridDim.x) {
    const uint8_t* srow = src + row * rowb;
    // rows start at arbitrary byte offsets: align the 16-byte loads on the global address, not on the row
    const int mis = (int)(reinterpret_cast<uintptr_t>(srow) & 15);
    const int head = mis ? 16 - mis : 0;
    for (int i = threadIdx.x; i < head && i < rowb; i += blockDim.x) rowbuf[i] = srow[i];
    const int nvec = (rowb - head) > 0 ? (rowb - head) / 16 : 0;
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
      const u32x4 x = *reinterpret_cast<const u32x4*>(srow + head + v * 16);
      // LDS destination head + 16v is 16-byte aligned only when head == 0: store as four dwords when it is not
      uint8_t* d = rowbuf + head + v * 16;
      if ((head & 3) == 0) {
        reinterpret_cast<unsigned*>(d)[0] = x[0];
        reinterpret_cast<unsigned*>(d)[1] = x[1];
        reinterpret_cast<unsigned*>(d)[2] = x[2];
        reinterpret_cast<unsigned*>(d)[3] = x[3];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) d[e] = (uint8_t)(x[e >> 2] >> (8 * (e & 3)));
      }
    }
    for (int i = head + nvec * 16 + threadIdx.x; i < rowb; i += blockDim.x) rowbuf[i] = srow[i];
    __syncthreads();
    uint8_t* drow = dst + row * dw * 3;
    for (int xo = threadIdx.x; xo < dw; xo += blockDim.x) {
      const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
      const int32_t* k = kk + (int64_t)xo * ksize;
      const uint8_t* s = rowbuf + xmin * 3;
      int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
      for (int x = 0; x < cnt; ++x) {
        const int c = k[x];
        a0 += (int)s[3 * x + 0] * c;
        a1 += (int)s[3 * x + 1] * c;
        a2 += (int)s[3 * x + 2] * c;
      }
      uint8_t* d = drow + xo * 3;
      if (swap_rb) {
        d[0] = clip8(a2);
        d[1] = clip8(a1);
        d[2] = clip8(a0);
      } else {
        d[0] = clip8(a0);
        d[1] = clip8(a1);
        d[2] = clip8(a2);
      }
    }
    __syncthreads();  // the next row overwrites rowbuf
  }
}

// thread = one output BYTE column element (x*3+c) of one output row: consecutive threads read consecutive bytes.
__global__ __launch_bounds__(256) void pil_resize_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                           int n, int sh, int dh, int w,
                                                           const int32_t* __restrict__ bounds,
                                                           const int32_t* __restrict__ kk, int ksize) {
  const int rowb = w * 3;
  const int64_t total = (int64_t)n * dh * rowb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xb = (int)(i % rowb);
    const int64_t r = i / rowb;
    const int yo = (int)(r % dh);
    const int img = (int)(r / dh);
    const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
    const int32_t* k = kk + (int64_t)yo * ksize;
    const uint8_t* s = src + ((int64_t)img * sh + ymin) * rowb + xb;
    int a = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) a += (int)s[(int64_t)y * rowb] * k[y];
    dst[i] = clip8(a);
  }
}

// the same pass with FOUR consecutive bytes per thread (row length a multiple of 4 bytes, 4-byte aligned images): one dword load
// per tap row and one dword store instead of four byte loads and four byte stores — identical integer arithmetic per byte
__global__ __launch_bounds__(256) void pil_resize_v4_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int sh,
                                                            int dh, int w, const int32_t* __restrict__ bounds,
                                                            const int32_t* __restrict__ kk, int ksize) {
  const int rowb = w * 3, roww = rowb / 4;
  const int64_t total = (int64_t)n * dh * roww;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xw = (int)(i % roww);
    const int64_t r = i / roww;
    const int yo = (int)(r % dh);
    const int img = (int)(r / dh);
    const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
    const int32_t* k = kk + (int64_t)yo * ksize;
    const uint8_t* s = src + ((int64_t)img * sh + ymin) * rowb + xw * 4;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
    for (int y = 0; y < cnt; ++y) {
      const unsigned v = *reinterpret_cast<const unsigned*>(s + (int64_t)y * rowb);
      const int c = k[y];
      a0 += (int)(v & 255u) * c;
      a1 += (int)((v >> 8) & 255u) * c;
      a2 += (int)((v >> 16) & 255u) * c;
      a3 += (int)(v >> 24) * c;
    }
    *reinterpret_cast<unsigned*>(dst + r * rowb + xw * 4) =
        (unsigned)clip8(a0) | ((unsigned)clip8(a1) << 8) | ((unsigned)clip8(a2) << 16) | ((unsigned)clip8(a3) << 24);
  }
}

// thread = one cropped pixel; writes its 3 normalised f16 values at the im2col position of its patch.
__global__ __launch_bounds__(256) void patchify_norm_kernel(const uint8_t* __restrict__ img, half_t* __restrict__ out,
                                                            int n, int ih, int iw, int top, int left, int gh, int gw,
                                                            int P, int64_t ldo, const float* __restrict__ lut) {
  const int ch = gh * P, cw = gw * P;
  const int64_t total = (int64_t)n * ch * cw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % cw);
    const int64_t r = i / cw;
    const int y = (int)(r % ch);
    const int b = (int)(r / ch);
    const uint8_t* s = img + (((int64_t)b * ih + top + y) * iw + left + x) * 3;
    const int py = y / P, ky = y - py * P, px = x / P, kx = x - px * P;
    half_t* d = out + (((int64_t)b * gh + py) * gw + px) * ldo + (ky * P + kx) * 3;
    // rescale + normalize depend only on (channel, byte): a 3x256 f32 table built on the host with the
    // processor's own numpy expression (lmx/resample.py) makes this step exact by construction.
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = (half_t)lut[c * 256 + s[c]];
  }
}

inline int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int lmx_k_pil_resize_h(const uint8_t* src, uint8_t* dst, int n, int sh, int sw, int dw, const int32_t* bounds,
                                  const int32_t* kk, int ksize, int swap_rb, lmx_stream_t stream) {
  LMX_REQUIRE(src && dst && bounds && kk, "lmx_k_pil_resize_h: null pointer");
  LMX_REQUIRE(n > 0 && sh > 0 && sw > 0 && dw > 0 && ksize > 0, "lmx_k_pil_resize_h: shape");
  LMX_REQUIRE(sw * 3 <= RESIZE_H_MAX_ROW, "lmx_k_pil_resize_h: source rows wider than %d pixels are not staged", RESIZE_H_MAX_ROW / 3);
  int64_t rows = (int64_t)n * sh;
  hipLaunchKernelGGL(pil_resize_h_kernel, dim3((unsigned)(rows < 256 * 16 ? rows : 256 * 16)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, dst, n, sh, sw, dw, bounds, kk, ksize, swap_rb);
  return lmx_launch_check("pil_resize_h_kernel");
}

extern "C" int lmx_k_pil_resize_v(const uint8_t* src, uint8_t* dst, int n, int sh, int dh, int w, const int32_t* bounds,
                                  const int32_t* kk, int ksize, lmx_stream_t stream) {
  LMX_REQUIRE(src && dst && bounds && kk, "lmx_k_pil_resize_v: null pointer");
  LMX_REQUIRE(n > 0 && sh > 0 && dh > 0 && w > 0 && ksize > 0, "lmx_k_pil_resize_v: shape");
  if ((w * 3) % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {  // four bytes per thread
    hipLaunchKernelGGL(pil_resize_v4_kernel, dim3(grid_for((int64_t)n * dh * (w * 3 / 4))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, dst, n, sh, dh, w, bounds, kk, ksize);
    return lmx_launch_check("pil_resize_v4_kernel");
  }
  hipLaunchKernelGGL(pil_resize_v_kernel, dim3(grid_for((int64_t)n * dh * w * 3)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, dst, n, sh, dh, w, bounds, kk, ksize);
  return lmx_launch_check("pil_resize_v_kernel");
}

extern "C" int lmx_k_patchify_norm(const uint8_t* img, void* out, int n, int ih, int iw, int top, int left, int gh,
                                   int gw, int P, int64_t ldo, const float* lut, lmx_stream_t stream) {
  LMX_REQUIRE(img && out && lut, "lmx_k_patchify_norm: null pointer");
  LMX_REQUIRE(n > 0 && gh > 0 && gw > 0 && P > 0 && ldo >= (int64_t)P * P * 3, "lmx_k_patchify_norm: shape");
  LMX_REQUIRE(top >= 0 && left >= 0 && top + gh * P <= ih && left + gw * P <= iw,
              "lmx_k_patchify_norm: crop (%d,%d)+(%d,%d) outside %dx%d", top, left, gh * P, gw * P, ih, iw);
  hipLaunchKernelGGL(patchify_norm_kernel, dim3(grid_for((int64_t)n * gh * P * gw * P)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), img, reinterpret_cast<half_t*>(out), n, ih, iw, top, left, gh,
                     gw, P, ldo, lut);
  return lmx_launch_check("patchify_norm_kernel");
}
