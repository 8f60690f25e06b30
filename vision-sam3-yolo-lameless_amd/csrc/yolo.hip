// yolo.hip — the YOLOv8 pieces that are not GEMM-shaped (SURVEY.md K1, K2-stem, K5, K6, K7 + scale_boxes).
// Everything here is HBM-bound byte / element work: one pass, 16-byte lane accesses along the NHWC channel axis.
// Replaces, under services/yolo-pipeline/app/main.py:76 (ultralytics, not in tree):
//   LetterBox (cv2.resize INTER_LINEAR + copyMakeBorder 114) + BGR->RGB + /255   -> letterbox_kernel + stem_conv
//   Conv(3->c, k3, s2)+BN+SiLU stem                                              -> stem_conv_kernel (VALU, f32)
//   SPPF max_pool2d(5,1,2)                                                       -> maxpool5_kernel
//   nn.Upsample(2,'nearest')                                                     -> upsample2_kernel
//   Detect: DFL softmax-expectation, dist2bbox, *stride, sigmoid                 -> detect_decode_kernel
//   ops.scale_boxes (pad removal, /gain, clip)                                   -> scale_boxes_kernel
#include "common.h"

namespace {

inline int grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- K1: cv2.resize(INTER_LINEAR) on u8 (fixed point, INTER_RESIZE_COEF_BITS = 11) fused with the 114 border
// and the BGR->RGB swap.  Tables come from the host (lmx/letterbox.py restates OpenCV's resizeGeneric_ table build):
// xofs[dw], ialpha[dw][2], yofs[dh], ibeta[dh][2].  One thread per output pixel:
//   H pass  : h(r) = S[r][sx]*a0 + S[r][sx+1]*a1                                  (int, scale 2^11)
//   V pass  : ((b0*(h(r0)>>4))>>16) + ((b1*(h(r1)>>4))>>16) + 2) >> 2            (VResizeLinear<uchar,...>)
// which is bit-identical to OpenCV's two-pass implementation.
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n,
                                                        int sh, int sw, int rh, int rw, int top, int left, int oh, int ow,
                                                        const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                        const int* __restrict__ yofs, const short* __restrict__ ibeta,
                                                        int swap_rb, int identity) {
  const int64_t total = (int64_t)n * oh * ow;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % ow);
    const int64_t r = i / ow;
    const int oy = (int)(r % oh);
    const int img = (int)(r / oh);
    uint8_t* d = dst + i * 3;
    const int dy = oy - top, dx = ox - left;
    if (dy < 0 || dy >= rh || dx < 0 || dx >= rw) {
      d[0] = d[1] = d[2] = 114;
      continue;
    }
    uint8_t v[3];
    const uint8_t* S = src + (int64_t)img * sh * sw * 3;
    if (identity) {  // LetterBox skips cv2.resize when the unpadded size equals the frame size
      const uint8_t* s = S + ((int64_t)dy * sw + dx) * 3;
      v[0] = s[0];
      v[1] = s[1];
      v[2] = s[2];
    } else {
      const int sx0 = xofs[dx];
      const int sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
      const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
      int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
      sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);
      sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
      const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
      const uint8_t* r0 = S + (int64_t)sy0 * sw * 3;
      const uint8_t* r1 = S + (int64_t)sy1 * sw * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[sx0 * 3 + c] * a0 + (int)r0[sx1 * 3 + c] * a1;
        const int h1 = (int)r1[sx0 * 3 + c] * a0 + (int)r1[sx1 * 3 + c] * a1;
        const int val = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v[c] = (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
      }
    }
    if (swap_rb) {
      d[0] = v[2];
      d[1] = v[1];
      d[2] = v[0];
    } else {
      d[0] = v[0];
      d[1] = v[1];
      d[2] = v[2];
    }
  }
}

// ---- stem: Conv(3 -> Cout, k3, s2, p1) + bias + SiLU on the u8 letterboxed frame; x = u8 / 255 in f32 (the
// predictor's `im.float() / 255`), f32 weights [ky][kx][c][Cout] staged in LDS, f32 FMA, f16 NHWC out.
// 0.2 GFLOP per 384x640 frame (0.2 % of the network): VALU is the right unit, the kernel is bound by its output.
template <int CO_T>  // output channels per thread pass
__global__ __launch_bounds__(256) void stem_conv_kernel(const uint8_t* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ bias, half_t* __restrict__ out, int n,
                                                        int H, int W, int Ho, int Wo, int Cout) {
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
    half_t* o = out + i * Cout;
    for (int c0 = 0; c0 < Cout; c0 += CO_T) {
      float acc[CO_T];
#pragma unroll
      for (int j = 0; j < CO_T; ++j) acc[j] = ws[27 * Cout + c0 + j];
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const float xv = x[t];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(xv, ws[t * Cout + c0 + j], acc[j]);
      }
      half8_t hv;
#pragma unroll
      for (int j = 0; j < CO_T; ++j) {
        const float v = acc[j] / (1.0f + expf(-acc[j]));
        hv[j] = (half_t)v;
      }
      *reinterpret_cast<half8_t*>(o + c0) = hv;
    }
  }
}

// ---- K5: max_pool2d(k=5, s=1, p=2) on an NHWC f16 channel slice -> another slice (pixel strides lds/ldd).
__global__ __launch_bounds__(256) void maxpool5_kernel(const half_t* __restrict__ src, int64_t lds, half_t* __restrict__ dst,
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
    half8_t m;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = (half_t)(-65504.0f);
    for (int dy = -2; dy <= 2; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = -2; dx <= 2; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const half8_t v = *reinterpret_cast<const half8_t*>(src + (((int64_t)b * H + yy) * W + xx) * lds + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    *reinterpret_cast<half8_t*>(dst + p * ldd + c) = m;
  }
}

// ---- K6: nearest x2 upsample of an NHWC f16 slice into a slice of the (2H x 2W) concat buffer.
__global__ __launch_bounds__(256) void upsample2_kernel(const half_t* __restrict__ src, int64_t lds, half_t* __restrict__ dst,
                                                        int64_t ldd, int n, int H, int W, int C) {
  const int cc = C / 8;
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)n * Ho * Wo * cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc) * 8;
    const int64_t p = i / cc;
    const int x = (int)(p % Wo);
    const int64_t r = p / Wo;
    const int y = (int)(r % Ho);
    const int b = (int)(r / Ho);
    *reinterpret_cast<half8_t*>(dst + p * ldd + c) =
        *reinterpret_cast<const half8_t*>(src + (((int64_t)b * H + (y >> 1)) * W + (x >> 1)) * lds + c);
  }
}

// ---- K7: Detect decode.  head f32 [n][H][W][64 + nc] per level (box logits: side*16 + bin, then class logits).
// One thread per (anchor, side) does the 16-bin softmax expectation; 4 neighbouring lanes exchange l,t,r,b with
// shuffles, lane side==0 writes the xywh; all 4 lanes share the nc sigmoids.  pred f32 [n][A][4+nc].
__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ head, float* __restrict__ pred, int n,
                                                            int H, int W, int nc, int ch, float stride, int a_off,
                                                            int A) {
  const int64_t total = (int64_t)n * H * W * 4;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < total; i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < total;
    const int side = (int)(i & 3);
    const int64_t cell = live ? (i >> 2) : 0;
    const int x = (int)(cell % W);
    const int64_t r = cell / W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const float* hrow = head + cell * ch;
    float l[16];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(hrow + side * 16 + k * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        l[k * 4 + e] = v[e];
        mx = fmaxf(mx, v[e]);
      }
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      l[k] = expf(l[k] - mx);
      se += l[k];
    }
    float dist = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) dist += (l[k] / se) * (float)k;  // softmax then the arange(16) 1x1 conv
    // gather (l, t, r, b) from the 4 lanes of this anchor
    const int base = threadIdx.x & ~3;
    const float dl = __shfl(dist, (base & 63) + 0, 64), dt = __shfl(dist, (base & 63) + 1, 64);
    const float dr = __shfl(dist, (base & 63) + 2, 64), db = __shfl(dist, (base & 63) + 3, 64);
    if (!live) continue;
    float* prow = pred + ((int64_t)b * A + a_off + (int64_t)y * W + x) * (4 + nc);
    if (side == 0) {
      const float ax = (float)x + 0.5f, ay = (float)y + 0.5f;
      const float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;
      prow[0] = ((x1 + x2) / 2.f) * stride;
      prow[1] = ((y1 + y2) / 2.f) * stride;
      prow[2] = (x2 - x1) * stride;
      prow[3] = (y2 - y1) * stride;
    }
    for (int c = side; c < nc; c += 4) {
      const float v = hrow[64 + c];
      prow[4 + c] = 1.0f / (1.0f + expf(-v));
    }
  }
}

// ---- scale_boxes: (xyxy - pad) / gain, clipped to the frame (ultralytics.utils.ops.scale_boxes + clip_boxes).
__global__ __launch_bounds__(256) void scale_boxes_kernel(float* __restrict__ boxes, int total, float padx, float pady,
                                                          float gain, float w, float h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float* b = boxes + (int64_t)i * 4;
  float x1 = (b[0] - padx) / gain, y1 = (b[1] - pady) / gain, x2 = (b[2] - padx) / gain, y2 = (b[3] - pady) / gain;
  b[0] = fminf(fmaxf(x1, 0.f), w);
  b[1] = fminf(fmaxf(y1, 0.f), h);
  b[2] = fminf(fmaxf(x2, 0.f), w);
  b[3] = fminf(fmaxf(y2, 0.f), h);
}

// ---- Pose head post-processing (ultralytics Pose.kpts_decode + ops.scale_coords + clip_coords) for the detections NMS
// kept: thread = (image, detection slot, keypoint).  The keypoints ride along by anchor index (`src` of lmx_k_nms), so only
// <= max_det anchors per image are decoded instead of all A.
struct PoseLevels {
  const float* raw[3];  // per level: f32 [n][h][w][ldk]
  int h[3], w[3], off[3];
  float stride[3];
};
__global__ __launch_bounds__(256) void pose_gather_kernel(const PoseLevels L, int64_t ldk, const int* __restrict__ src,
                                                          const int* __restrict__ counts, int n, int max_det, int K, int ndim,
                                                          float padx, float pady, float gain, float fw, float fh,
                                                          float* __restrict__ out) {
  const int64_t total = (int64_t)n * max_det * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int64_t r = i / K;
    const int j = (int)(r % max_det);
    const int b = (int)(r / max_det);
    float* o = out + i * ndim;
    if (j >= counts[b]) {
      for (int e = 0; e < ndim; ++e) o[e] = 0.f;
      continue;
    }
    const int a = src[(int64_t)b * max_det + j];
    const int l = a >= L.off[2] ? 2 : (a >= L.off[1] ? 1 : 0);
    const int local = a - L.off[l];
    const int y = local / L.w[l], x = local - y * L.w[l];
    const float* v = L.raw[l] + (((int64_t)b * L.h[l] + y) * L.w[l] + x) * ldk + k * ndim;
    // kpts_decode: (v * 2 + (anchor - 0.5)) * stride with anchor = cell + 0.5; then scale_coords: (c - pad) / gain, clipped
    const float X = (v[0] * 2.0f + (float)x) * L.stride[l];
    const float Y = (v[1] * 2.0f + (float)y) * L.stride[l];
    o[0] = fminf(fmaxf((X - padx) / gain, 0.f), fw);
    o[1] = fminf(fmaxf((Y - pady) / gain, 0.f), fh);
    if (ndim == 3) o[2] = 1.0f / (1.0f + expf(-v[2]));
  }
}

}  // namespace

extern "C" int lmx_k_pose_gather(const float* raw0, const float* raw1, const float* raw2, int64_t ldk, const int32_t* hw,
                                 const float* strides, const int32_t* src, const int32_t* counts, int n, int max_det, int K,
                                 int ndim, float padx, float pady, float gain, float w, float h, float* out,
                                 lmx_stream_t stream) {
  LMX_REQUIRE(raw0 && raw1 && raw2 && hw && strides && src && counts && out, "lmx_k_pose_gather: null pointer");
  LMX_REQUIRE(n > 0 && max_det > 0 && K > 0 && (ndim == 2 || ndim == 3) && ldk >= (int64_t)K * ndim && gain > 0.f,
              "lmx_k_pose_gather: arguments");
  PoseLevels L;
  L.raw[0] = raw0;
  L.raw[1] = raw1;
  L.raw[2] = raw2;
  int off = 0;
  for (int l = 0; l < 3; ++l) {
    LMX_REQUIRE(hw[2 * l] > 0 && hw[2 * l + 1] > 0, "lmx_k_pose_gather: level %d geometry", l);
    L.h[l] = hw[2 * l];
    L.w[l] = hw[2 * l + 1];
    L.off[l] = off;
    L.stride[l] = strides[l];
    off += L.h[l] * L.w[l];
  }
  hipLaunchKernelGGL(pose_gather_kernel, dim3(grid_for((int64_t)n * max_det * K)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), L, ldk, src, counts, n, max_det, K, ndim, padx, pady, gain, w, h, out);
  return lmx_launch_check("pose_gather_kernel");
}

extern "C" int lmx_k_letterbox(const uint8_t* src, uint8_t* dst, int n, int sh, int sw, int rh, int rw, int top, int left,
                               int oh, int ow, const int32_t* xofs, const int16_t* ialpha, const int32_t* yofs,
                               const int16_t* ibeta, int swap_rb, lmx_stream_t stream) {
  LMX_REQUIRE(src && dst, "lmx_k_letterbox: null pointer");
  LMX_REQUIRE(n > 0 && sh > 0 && sw > 0 && rh > 0 && rw > 0 && oh >= top + rh && ow >= left + rw && top >= 0 && left >= 0,
              "lmx_k_letterbox: geometry");
  const int identity = (rh == sh && rw == sw) ? 1 : 0;
  LMX_REQUIRE(identity || (xofs && ialpha && yofs && ibeta), "lmx_k_letterbox: null tables");
  hipLaunchKernelGGL(letterbox_kernel, dim3(grid_for((int64_t)n * oh * ow)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, dst, n, sh, sw, rh, rw, top, left, oh, ow, xofs, ialpha, yofs,
                     ibeta, swap_rb, identity);
  return lmx_launch_check("letterbox_kernel");
}

extern "C" int lmx_k_stem_conv(const uint8_t* img, const float* w, const float* bias, void* out, int n, int H, int W, int Cout,
                               lmx_stream_t stream) {
  LMX_REQUIRE(img && w && bias && out, "lmx_k_stem_conv: null pointer");
  LMX_REQUIRE(n > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && Cout <= 256, "lmx_k_stem_conv: shape (Cout %d)", Cout);
  LMX_REQUIRE(aligned16(out), "lmx_k_stem_conv: out alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t smem = (size_t)28 * Cout * sizeof(float);
  hipLaunchKernelGGL((stem_conv_kernel<8>), dim3(grid_for((int64_t)n * Ho * Wo)), dim3(256), smem,
                     reinterpret_cast<hipStream_t>(stream), img, w, bias, reinterpret_cast<half_t*>(out), n, H, W, Ho, Wo, Cout);
  return lmx_launch_check("stem_conv_kernel");
}

extern "C" int lmx_k_maxpool5(const void* src, int64_t lds, void* dst, int64_t ldd, int n, int H, int W, int C,
                              lmx_stream_t stream) {
  LMX_REQUIRE(src && dst, "lmx_k_maxpool5: null pointer");
  LMX_REQUIRE(n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && lds >= C && ldd >= C,
              "lmx_k_maxpool5: shape");
  LMX_REQUIRE(aligned16(src) && aligned16(dst), "lmx_k_maxpool5: alignment");
  hipLaunchKernelGGL(maxpool5_kernel, dim3(grid_for((int64_t)n * H * W * (C / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const half_t*>(src), lds,
                     reinterpret_cast<half_t*>(dst), ldd, n, H, W, C);
  return lmx_launch_check("maxpool5_kernel");
}

extern "C" int lmx_k_upsample2(const void* src, int64_t lds, void* dst, int64_t ldd, int n, int H, int W, int C,
                               lmx_stream_t stream) {
  LMX_REQUIRE(src && dst, "lmx_k_upsample2: null pointer");
  LMX_REQUIRE(n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && lds >= C && ldd >= C,
              "lmx_k_upsample2: shape");
  LMX_REQUIRE(aligned16(src) && aligned16(dst), "lmx_k_upsample2: alignment");
  hipLaunchKernelGGL(upsample2_kernel, dim3(grid_for((int64_t)n * 4 * H * W * (C / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const half_t*>(src), lds,
                     reinterpret_cast<half_t*>(dst), ldd, n, H, W, C);
  return lmx_launch_check("upsample2_kernel");
}

extern "C" int lmx_k_detect_decode(const float* head, int64_t ldh, float* pred, int n, int H, int W, int nc, float stride,
                                   int a_off, int A, lmx_stream_t stream) {
  LMX_REQUIRE(head && pred, "lmx_k_detect_decode: null pointer");
  LMX_REQUIRE(n > 0 && H > 0 && W > 0 && nc > 0 && a_off >= 0 && a_off + H * W <= A, "lmx_k_detect_decode: shape");
  LMX_REQUIRE(ldh >= 64 + nc && ldh % 4 == 0 && aligned16(head), "lmx_k_detect_decode: head row stride %lld", (long long)ldh);
  hipLaunchKernelGGL(detect_decode_kernel, dim3(grid_for((int64_t)n * H * W * 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), head, pred, n, H, W, nc, (int)ldh, stride, a_off, A);
  return lmx_launch_check("detect_decode_kernel");
}

extern "C" int lmx_k_scale_boxes(float* boxes, int total, float padx, float pady, float gain, float w, float h,
                                 lmx_stream_t stream) {
  LMX_REQUIRE(boxes && total > 0 && gain > 0.f, "lmx_k_scale_boxes: arguments");
  hipLaunchKernelGGL(scale_boxes_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), boxes,
                     total, padx, pady, gain, w, h);
  return lmx_launch_check("scale_boxes_kernel");
}
