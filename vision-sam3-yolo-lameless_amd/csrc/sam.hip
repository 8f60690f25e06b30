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
// KW_T > 0: the kernel width as a compile-time constant (7: Hiera's patch embedding, 16: the SAM ViT's) — the two divisions per
// VALUE by run-time widths made this kernel ALU-bound at 1.3 TB/s of output; the look-up table sits in LDS.
template <int KW_T>
__global__ __launch_bounds__(256) void im2col_u8_kernel(const uint8_t* __restrict__ img, const float* __restrict__ lut,
                                                        half_t* __restrict__ out, int n, int rh, int rw, int OH, int OW, int KH,
                                                        int KW_rt, int stride, int pad, int64_t ldo) {
  __shared__ float lut_s[768];
  for (int i = threadIdx.x; i < 768; i += blockDim.x) lut_s[i] = lut[i];
  __syncthreads();
  const int KW = KW_T > 0 ? KW_T : KW_rt;
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
    const uint8_t* ib = img + (int64_t)b * rh * rw * 3;
    const int y0 = oy * stride - pad, x0 = ox * stride - pad;
    half8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ck * 8 + e;
      float val = 0.f;
      if (k < K) {
        const int tap = k / 3, c = k - tap * 3;
        const int ky = tap / KW, kx = tap - ky * KW;
        const int y = y0 + ky, x = x0 + kx;
        if ((unsigned)y < (unsigned)rh && (unsigned)x < (unsigned)rw) val = lut_s[c * 256 + ib[(y * rw + x) * 3 + c]];
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


// out[r] = a[r] + b[r % b_rows]
template <int IN_DT, int OUT_DT>
__global__ __launch_bounds__(256) void add_bcast_kernel(const void* __restrict__ av, int64_t lda, const float* __restrict__ b,
                                                        int64_t ldb, int b_rows, void* __restrict__ out, int64_t ldo, int64_t rows,
                                                        int D) {
  const int cc = D / 4;
  const int64_t total = rows * cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc) * 4;
    const int64_t r = i / cc;
    f32x4 v;
    if (IN_DT == LMX_F32) {
      v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(av) + r * lda + c);
    } else {
      const half4_t hv = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(av) + r * lda + c);
      v = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
    }
    v += *reinterpret_cast<const f32x4*>(b + (r % b_rows) * ldb + c);
    if (OUT_DT == LMX_F32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + r * ldo + c) = v;
    } else {
      half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(out) + r * ldo + c) = h;
    }
  }
}

// thread = one of the 16 sub-pixels of one grid cell: dot(up[..][C], hyper[n][C]) written at its spatial position.
__global__ __launch_bounds__(256) void hyper_mask_kernel(const half_t* __restrict__ up, const float* __restrict__ hyper,
                                                         float* __restrict__ logits, int n, int G, int C) {
  const int64_t total = (int64_t)n * G * G * 16;
  const int S = 4 * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q2 = (int)(i & 3), q1 = (int)((i >> 2) & 3);
    const int64_t cell = i >> 4;
    const int x = (int)(cell % G);
    const int64_t r = cell / G;
    const int y = (int)(r % G);
    const int b = (int)(r / G);
    const half_t* u = up + i * C;
    const float* hy = hyper + (int64_t)b * C;
    float acc = 0.f;
    for (int c = 0; c < C; c += 8) {
      const half8_t v = *reinterpret_cast<const half8_t*>(u + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf((float)v[e], hy[c + e], acc);
    }
    const int Y = 4 * y + 2 * (q1 >> 1) + (q2 >> 1), X = 4 * x + 2 * (q1 & 1) + (q2 & 1);
    logits[((int64_t)b * S + Y) * S + X] = acc;
  }
}

// torch upsample_bilinear2d(align_corners=False) source index: scale*(dst+0.5)-0.5 clamped at 0
__device__ __forceinline__ void bil_idx(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__device__ __forceinline__ float sample_mid(const float* __restrict__ lg, int L, int T, float sLT, int Y, int X) {
  // value of the TxT bilinear upsample of the LxL logits at integer position (Y, X)
  int y0, y1, x0, x1;
  float ly, lx;
  bil_idx(sLT, Y, L, y0, y1, ly);
  bil_idx(sLT, X, L, x0, x1, lx);
  const float t0 = (1.f - lx) * lg[y0 * L + x0] + lx * lg[y0 * L + x1];
  const float t1 = (1.f - lx) * lg[y1 * L + x0] + lx * lg[y1 * L + x1];
  return (1.f - ly) * t0 + ly * t1;
}

// pass 1: the T x T bilinear upsample of the L x L logits, only the [nh][nw] crop that pass 2 reads
// DBG 0 is the product (plain loads).  LMX_DBG_MASK selects the others in the development build only (make dbg, see
// ld_mid): 1 = as the product but compiled WITH the SLP vectoriser in liblmx_dbg.so (reproduces the defect); 2 = pass 1
// counts its finished workgroups in stats[7] and pass 2 counts in stats[15] the workgroups that started early (never
// seen); 3 / 4 / 5 = system-scope release at the end of pass 1 / acquire at the start of pass 2 / both; 6 / 7 / 9 =
// agent- / workgroup- / system-scope atomic loads in pass 2; 8 = s_waitcnt vmcnt(0) after every pixel's four loads
template <int DBG>
__global__ __launch_bounds__(256) void mask_mid_kernel(const float* __restrict__ logits, float* __restrict__ mid, int n, int L,
                                                       int T, int nh, int nw, unsigned long long* dbg) {
  const float sLT = (float)L / (float)T;
  const int64_t total = (int64_t)n * nh * nw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % nw);
    const int64_t r = i / nw;
    const int Y = (int)(r % nh);
    const int b = (int)(r / nh);
    mid[i] = sample_mid(logits + (int64_t)b * L * L, L, T, sLT, Y, X);
  }
  if (DBG == 2) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&dbg[7], 1ull);
  }
  if (DBG == 3 || DBG == 5) __atomic_thread_fence(__ATOMIC_RELEASE);  // system-scope release (L2 write-back) by every wave of pass 1
}

// How pass 2 loads the intermediate.  Round 1 saw pass 2 produce a few wrong pixels per launch with PLAIN loads whenever
// attn_kernel<QB=2> workgroups were co-resident, and never with atomic loads of any scope - and blamed the loads.  Round 2
// found the cause in the arithmetic: with plain loads hipcc's SLP vectoriser packed the bilinear
// taps into `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`, a form that misreads src1's high register on gfx950 while
// another wave of the SIMD issues MFMAs; with atomic loads the vectoriser happened to choose the src0 form, which works.
// The product (DBG 0) therefore uses plain loads again and the build forbids the instruction form (tools/isa_lint.py,
// -fno-slp-vectorize).  The other variants exist only in the development build (make dbg).
template <int DBG>
__device__ __forceinline__ float ld_mid(const float* p) {
  if (DBG == 9) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (DBG == 6) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (DBG == 7) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return *p;
}

// pass 2: bilinear [nh][nw] -> [h][w], > 0, statistics.  A thread produces 4 consecutive pixels of one row (row
// interpolation weights computed once, one 4-byte store instead of four byte stores).
template <int DBG>
__global__ __launch_bounds__(256) void mask_post_kernel(const float* __restrict__ mid, int n, int nh, int nw, int h, int w,
                                                        uint8_t* __restrict__ mask, unsigned long long* __restrict__ stats,
                                                        unsigned expect) {
  const int b = blockIdx.y;
  if (DBG == 4 || DBG == 5) __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system-scope acquire (cache invalidate) by every wave of pass 2
  if (DBG == 2 && threadIdx.x == 0) {
    const unsigned long long c = __hip_atomic_load(&stats[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c < expect) atomicAdd(&stats[15], 1ull);
  }
  const float* md = mid + (int64_t)b * nh * nw;
  const float sy = (float)nh / (float)h, sx = (float)nw / (float)w;
  unsigned long long area = 0, sumx = 0, sumy = 0;
  int minx = 0x7fffffff, miny = 0x7fffffff, maxx = -1, maxy = -1;
  const int wq = (w + 3) / 4;
  const int total = h * wq;  // < 2^31 (checked by the launcher)
  uint8_t* mk = mask + (int64_t)b * h * w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int y = i / wq, xq = (i - y * wq) * 4;
    int Y0, Y1;
    float ly;
    bil_idx(sy, y, nh, Y0, Y1, ly);
    const float* r0 = md + (int64_t)Y0 * nw;
    const float* r1 = md + (int64_t)Y1 * nw;
    unsigned packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = xq + e;
      if (x >= w) break;
      int X0, X1;
      float lx;
      bil_idx(sx, x, nw, X0, X1, lx);
      const float a00 = ld_mid<DBG>(r0 + X0), a01 = ld_mid<DBG>(r0 + X1), a10 = ld_mid<DBG>(r1 + X0), a11 = ld_mid<DBG>(r1 + X1);
      if (DBG == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // plain loads, but never a partial vmcnt wait
      const float t0 = (1.f - lx) * a00 + lx * a01;
      const float t1 = (1.f - lx) * a10 + lx * a11;
      const float v = (1.f - ly) * t0 + ly * t1;
      if (v > 0.0f) {
        packed |= 1u << (8 * e);
        ++area;
        sumx += (unsigned)x;
        sumy += (unsigned)y;
        minx = x < minx ? x : minx;
        maxx = x > maxx ? x : maxx;
        miny = y < miny ? y : miny;
        maxy = y > maxy ? y : maxy;
      }
    }
    if (xq + 4 <= w && (w & 3) == 0) {
      *reinterpret_cast<unsigned*>(mk + (int64_t)y * w + xq) = packed;
    } else {
      for (int e = 0; e < 4 && xq + e < w; ++e) mk[(int64_t)y * w + xq + e] = (packed >> (8 * e)) & 1;
    }
  }
  // wave reduction, then block reduction through LDS, then ONE set of atomics per block: same-address 64-bit atomics
  // serialise at the memory side, so their count (blocks per frame x 7), not the pixel work, used to set this kernel's time
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    area += __shfl_xor(area, o, 64);
    sumx += __shfl_xor(sumx, o, 64);
    sumy += __shfl_xor(sumy, o, 64);
    const int a1 = __shfl_xor(minx, o, 64), a2 = __shfl_xor(miny, o, 64), a3 = __shfl_xor(maxx, o, 64), a4 = __shfl_xor(maxy, o, 64);
    minx = a1 < minx ? a1 : minx;
    miny = a2 < miny ? a2 : miny;
    maxx = a3 > maxx ? a3 : maxx;
    maxy = a4 > maxy ? a4 : maxy;
  }
  __shared__ unsigned long long red_s[4][3];
  __shared__ int red_b[4][4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red_s[wv][0] = area;
    red_s[wv][1] = sumx;
    red_s[wv][2] = sumy;
    red_b[wv][0] = minx;
    red_b[wv][1] = miny;
    red_b[wv][2] = maxx;
    red_b[wv][3] = maxy;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      area += red_s[k][0];
      sumx += red_s[k][1];
      sumy += red_s[k][2];
      minx = red_b[k][0] < minx ? red_b[k][0] : minx;
      miny = red_b[k][1] < miny ? red_b[k][1] : miny;
      maxx = red_b[k][2] > maxx ? red_b[k][2] : maxx;
      maxy = red_b[k][3] > maxy ? red_b[k][3] : maxy;
    }
    if (area) {
      unsigned long long* st = stats + (int64_t)b * 8;
      atomicAdd(&st[0], area);
      atomicAdd(&st[1], sumx);
      atomicAdd(&st[2], sumy);
      atomicMin(reinterpret_cast<long long*>(&st[3]), (long long)minx);
      atomicMin(reinterpret_cast<long long*>(&st[4]), (long long)miny);
      atomicMax(reinterpret_cast<long long*>(&st[5]), (long long)maxx);
      atomicMax(reinterpret_cast<long long*>(&st[6]), (long long)maxy);
    }
  }
}

// thread = one output byte = 8 mask pixels, first pixel in the most significant bit (numpy.packbits order).  For 0/1
// bytes loaded little-endian as one 64-bit word, (x * 0x8040201008040201) >> 56 gathers the eight low bits in that order.
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t* __restrict__ src, int64_t rows, int w, int wb,
                                                        uint8_t* __restrict__ dst) {
  const int64_t total = rows * wb;
  const bool fast = (w & 7) == 0 && ((reinterpret_cast<uintptr_t>(src) & 7) == 0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / wb;
    const int c = (int)(i - r * wb);
    const uint8_t* sp = src + r * w + (int64_t)c * 8;
    unsigned long long x = 0;
    if (fast) {
      x = *reinterpret_cast<const unsigned long long*>(sp);
    } else {
      for (int e = 0; e < 8; ++e)
        if (c * 8 + e < w) x |= (unsigned long long)sp[e] << (8 * e);
    }
    x = (x | (x >> 1) | (x >> 2) | (x >> 3) | (x >> 4) | (x >> 5) | (x >> 6) | (x >> 7)) & 0x0101010101010101ull;  // any non-zero byte -> 1
    dst[i] = (uint8_t)((x * 0x8040201008040201ull) >> 56);
  }
}

__global__ __launch_bounds__(256) void prompt_box_kernel(const float* __restrict__ boxes, int64_t ldb, float* __restrict__ sparse,
                                                        int n, double sx, double sy, float S, const float* __restrict__ gauss,
                                                        const float* __restrict__ corner, int F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (frame, corner, j)
  if (i >= n * 2 * F) return;
  const int j = i % F, k = (i / F) & 1, b = i / (2 * F);
  const float* bx = boxes + (int64_t)b * ldb + 2 * k;
  // apply_boxes in double, then the f32 tensor; + 0.5 (pixel centre), / S, 2c - 1
  const float px = (float)((double)bx[0] * sx) + 0.5f, py = (float)((double)bx[1] * sy) + 0.5f;
  const float cx = 2.f * (px / S) - 1.f, cy = 2.f * (py / S) - 1.f;
  const float ang = 6.283185307179586f * (cx * gauss[j] + cy * gauss[F + j]);
  float* o = sparse + ((int64_t)b * 2 + k) * 2 * F;
  o[j] = sinf(ang) + corner[k * 2 * F + j];
  o[F + j] = cosf(ang) + corner[k * 2 * F + F + j];
}

__global__ void mask_stats_init_kernel(long long* stats, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 8) return;
  const int f = i & 7;
  stats[i] = (f == 3 || f == 4) ? 0x7fffffffll : ((f == 5 || f == 6) ? -1ll : 0ll);
}

}  // namespace

extern "C" int lmx_k_im2col_u8(const uint8_t* img, const float* lut, void* out, int n, int rh, int rw, int IH, int IW, int KH,
                               int KW, int stride, int pad, int64_t ldo, lmx_stream_t stream) {
  LMX_REQUIRE(img && lut && out, "lmx_k_im2col_u8: null pointer");
  LMX_REQUIRE(n > 0 && rh > 0 && rw > 0 && rh <= IH && rw <= IW && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
              "lmx_k_im2col_u8: geometry");
  LMX_REQUIRE(ldo % 8 == 0 && ldo >= (int64_t)KH * KW * 3 && aligned16(out), "lmx_k_im2col_u8: ldo/alignment");
  const int OH = (IH + 2 * pad - KH) / stride + 1, OW = (IW + 2 * pad - KW) / stride + 1;
  LMX_REQUIRE((int64_t)rh * rw * 3 < 0x7fffffffll, "lmx_k_im2col_u8: frame too large");
  const dim3 grid(grid_for((int64_t)n * OH * OW * (ldo / 8)));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  half_t* o16 = reinterpret_cast<half_t*>(out);
  if (KW == 7)
    hipLaunchKernelGGL((im2col_u8_kernel<7>), grid, dim3(256), 0, st, img, lut, o16, n, rh, rw, OH, OW, KH, KW, stride, pad, ldo);
  else if (KW == 16)
    hipLaunchKernelGGL((im2col_u8_kernel<16>), grid, dim3(256), 0, st, img, lut, o16, n, rh, rw, OH, OW, KH, KW, stride, pad, ldo);
  else
    hipLaunchKernelGGL((im2col_u8_kernel<0>), grid, dim3(256), 0, st, img, lut, o16, n, rh, rw, OH, OW, KH, KW, stride, pad, ldo);
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

extern "C" int lmx_k_prompt_box(const float* boxes, int64_t ldb, float* sparse, int n, double sx, double sy, float S,
                                const float* gauss, const float* corner, int F, lmx_stream_t stream) {
  LMX_REQUIRE(boxes && sparse && gauss && corner && n > 0 && F > 0 && ldb >= 4 && S > 0.f, "lmx_k_prompt_box: arguments");
  hipLaunchKernelGGL(prompt_box_kernel, dim3((n * 2 * F + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), boxes, ldb,
                     sparse, n, sx, sy, S, gauss, corner, F);
  return lmx_launch_check("prompt_box_kernel");
}

extern "C" int lmx_k_add_bcast(const void* a, int a_dtype, int64_t lda, const float* b, int64_t ldb, int b_rows, void* out,
                               int out_dtype, int64_t ldo, int64_t rows, int D, lmx_stream_t stream) {
  LMX_REQUIRE(a && b && out, "lmx_k_add_bcast: null pointer");
  LMX_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && b_rows > 0 && lda >= D && ldb >= D && ldo >= D && lda % 4 == 0 && ldb % 4 == 0 &&
                  ldo % 4 == 0, "lmx_k_add_bcast: shape");
  LMX_REQUIRE(((((uintptr_t)a) & (a_dtype == LMX_F32 ? 15 : 7)) == 0) && aligned16(b) &&
                  ((((uintptr_t)out) & (out_dtype == LMX_F32 ? 15 : 7)) == 0), "lmx_k_add_bcast: alignment");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for(rows * (D / 4));
  if (a_dtype == LMX_F32 && out_dtype == LMX_F32)
    hipLaunchKernelGGL((add_bcast_kernel<LMX_F32, LMX_F32>), dim3(g), dim3(256), 0, st, a, lda, b, ldb, b_rows, out, ldo, rows, D);
  else if (a_dtype == LMX_F32 && out_dtype == LMX_F16)
    hipLaunchKernelGGL((add_bcast_kernel<LMX_F32, LMX_F16>), dim3(g), dim3(256), 0, st, a, lda, b, ldb, b_rows, out, ldo, rows, D);
  else if (a_dtype == LMX_F16 && out_dtype == LMX_F32)
    hipLaunchKernelGGL((add_bcast_kernel<LMX_F16, LMX_F32>), dim3(g), dim3(256), 0, st, a, lda, b, ldb, b_rows, out, ldo, rows, D);
  else if (a_dtype == LMX_F16 && out_dtype == LMX_F16)
    hipLaunchKernelGGL((add_bcast_kernel<LMX_F16, LMX_F16>), dim3(g), dim3(256), 0, st, a, lda, b, ldb, b_rows, out, ldo, rows, D);
  else
    LMX_REQUIRE(false, "lmx_k_add_bcast: dtypes %d -> %d", a_dtype, out_dtype);
  return lmx_launch_check("add_bcast_kernel");
}

extern "C" int lmx_k_hyper_mask(const void* up, const float* hyper, float* logits, int n, int G, int C, lmx_stream_t stream) {
  LMX_REQUIRE(up && hyper && logits, "lmx_k_hyper_mask: null pointer");
  LMX_REQUIRE(n > 0 && G > 0 && C > 0 && C % 8 == 0 && C <= 64 && aligned16(up), "lmx_k_hyper_mask: shape");
  hipLaunchKernelGGL(hyper_mask_kernel, dim3(grid_for((int64_t)n * G * G * 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const half_t*>(up), hyper, logits, n, G, C);
  return lmx_launch_check("hyper_mask_kernel");
}

extern "C" int lmx_k_mask_post(const float* logits, int n, int L, int T, int nh, int nw, int h, int w, uint8_t* mask, int64_t* stats,
                               float* workspace, lmx_stream_t stream) {
  LMX_REQUIRE(logits && mask && stats && workspace, "lmx_k_mask_post: null pointer");
  LMX_REQUIRE(n > 0 && L > 0 && T >= L && nh > 0 && nw > 0 && nh <= T && nw <= T && h > 0 && w > 0 &&
                  (int64_t)h * ((w + 3) / 4) < 0x7fffffffll - 96 * 256 && n <= 65535, "lmx_k_mask_post: geometry");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mask_stats_init_kernel, dim3((n * 8 + 255) / 256), dim3(256), 0, st, reinterpret_cast<long long*>(stats), n);
  int gx = (int)(((int64_t)h * ((w + 3) / 4) + 255) / 256);
  if (gx > 96) gx = 96;  // 96 blocks x n frames: >= 3000 blocks at the bench batch, and only 96 x 7 atomics per frame
  const unsigned gmid = grid_for((int64_t)n * nh * nw);
  unsigned long long* su = reinterpret_cast<unsigned long long*>(stats);
#ifdef LMX_DBG_VARIANTS  // development build only (make dbg -> liblmx_dbg.so): the probes of DESIGN.md section 6
  static const int dbg = getenv("LMX_DBG_MASK") ? atoi(getenv("LMX_DBG_MASK")) : 0;
#define LMX_MASK_VARIANT(M, P)                                                                                         \
  {                                                                                                                    \
    hipLaunchKernelGGL(mask_mid_kernel<M>, dim3(gmid), dim3(256), 0, st, logits, workspace, n, L, T, nh, nw, su);      \
    hipLaunchKernelGGL(mask_post_kernel<P>, dim3(gx, n), dim3(256), 0, st, workspace, n, nh, nw, h, w, mask, su, gmid); \
  }
  if (dbg == 2 && n >= 2) LMX_MASK_VARIANT(2, 2)
  else if (dbg == 3) LMX_MASK_VARIANT(3, 3)  // plain loads + release at the end of pass 1
  else if (dbg == 4) LMX_MASK_VARIANT(4, 4)  // plain loads + acquire at the start of pass 2
  else if (dbg == 5) LMX_MASK_VARIANT(5, 5)  // plain loads + both fences
  else if (dbg == 8) LMX_MASK_VARIANT(1, 8)
  else if (dbg == 6) LMX_MASK_VARIANT(1, 6)
  else if (dbg == 7) LMX_MASK_VARIANT(1, 7)
  else if (dbg == 1) LMX_MASK_VARIANT(1, 1)
  else if (dbg == 9) LMX_MASK_VARIANT(1, 9)  // system-scope loads (the round-1 product)
  else LMX_MASK_VARIANT(0, 0)
#undef LMX_MASK_VARIANT
#else
  hipLaunchKernelGGL(mask_mid_kernel<0>, dim3(gmid), dim3(256), 0, st, logits, workspace, n, L, T, nh, nw, su);
  hipLaunchKernelGGL(mask_post_kernel<0>, dim3(gx, n), dim3(256), 0, st, workspace, n, nh, nw, h, w, mask, su, gmid);
#endif
  return lmx_launch_check("mask_post_kernel");
}

extern "C" int lmx_k_pack_bits(const uint8_t* src, int64_t rows, int w, uint8_t* dst, lmx_stream_t stream) {
  LMX_REQUIRE(src && dst, "lmx_k_pack_bits: null pointer");
  LMX_REQUIRE(rows > 0 && w > 0, "lmx_k_pack_bits: rows=%lld w=%d", (long long)rows, w);
  const int wb = (w + 7) / 8;
  hipLaunchKernelGGL(pack_bits_kernel, dim3(grid_for(rows * wb)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, rows, w, wb, dst);
  return lmx_launch_check("pack_bits_kernel");
}
