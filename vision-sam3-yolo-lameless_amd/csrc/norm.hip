// norm.hip — LayerNorm (K11), token assembly, token mean-pool (K22), DINOv3 RoPE.
// All HBM-bound: one pass over the data, 16-byte lane accesses, wave-shuffle reductions, no LDS.
#include "common.h"
#include <stdlib.h>

namespace {

// One 64-lane wave per row; the row is held in registers (<= 16 float4 per lane) between the mean pass and the
// variance pass, so HBM is read exactly once.  Matches torch.nn.LayerNorm (biased variance, eps inside rsqrt).
// Narrow rows (D <= 128, Hiera stage 1): a 64-lane wave would leave more than half its lanes idle, so a row is given to
// a 32-lane half-wave (LPR = 32) and the reductions stop at xor 16.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int IN_DT, int OUT_DT>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const void* __restrict__ xv, int64_t ldx,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, void* __restrict__ yv,
                                                               int64_t ldy, int rows, int D, float eps, int act) {
  const int l = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool live = row < rows;
  const int c = l * 4;
  const bool ok = live && c < D;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    if (IN_DT == LMX_F32) {
      v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xv) + (int64_t)row * ldx + c);
    } else {
      const half4_t hv = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(xv) + (int64_t)row * ldx + c);
      v = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
    }
  }
  const float mean = group_sum<32>((v[0] + v[1]) + (v[2] + v[3])) / (float)D;
  const f32x4 dl = ok ? v - mean : f32x4{0.f, 0.f, 0.f, 0.f};
  const float var = group_sum<32>((dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3])) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (!ok) return;
  f32x4 o = dl * rstd * *reinterpret_cast<const f32x4*>(gamma + c) + *reinterpret_cast<const f32x4*>(beta + c);
  if (act == LMX_ACT_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = 0.5f * o[e] * (1.0f + erff(o[e] * 0.70710678118654752440f));
  }
  if (OUT_DT == LMX_F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(yv) + (int64_t)row * ldy + c) = o;
  } else {
    half4_t h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
    *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(yv) + (int64_t)row * ldy + c) = h;
  }
}

template <int IN_DT, int OUT_DT, int ITERS>  // ITERS float4 per lane: D <= 256*ITERS (fewer registers -> more rows in flight)
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ xv, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ yv,
                                                        int64_t ldy, int rows, int D, float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[ITERS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      if (IN_DT == LMX_F32) {
        v[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xv) + (int64_t)row * ldx + c);
      } else {
        const half4_t hv = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(xv) + (int64_t)row * ldx + c);
        v[i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const f32x4 dlt = v[i] - mean;
      q += (dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o = (v[i] - mean) * rstd * g + b;
      if (act == LMX_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = 0.5f * o[e] * (1.0f + erff(o[e] * 0.70710678118654752440f));
      }
      if (OUT_DT == LMX_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(yv) + (int64_t)row * ldy + c) = o;
      } else {
        half4_t h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
        *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(yv) + (int64_t)row * ldy + c) = h;
      }
    }
  }
}

// the same rows, several per wave: a wave walks rows w, w + W, ... and requests row r + W before it reduces row r (one row per
// short-lived wave leaves the memory system waiting on wave launches: 4.7 TB/s on the f32 -> f16 LayerNorm of the ViT
// blocks).  Same per-row arithmetic as layernorm_kernel: identical bits.
template <int IN_DT, int OUT_DT, int ITERS>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const void* __restrict__ xv, int64_t ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, void* __restrict__ yv, int64_t ldy, int rows,
                                                             int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 g[ITERS], b[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int c = (i * 64 + lane) * 4;
    g[i] = c < D ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    b[i] = c < D ? *reinterpret_cast<const f32x4*>(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto load = [&](int r, f32x4* v) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c = (i * 64 + lane) * 4;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < D) {
        if (IN_DT == LMX_F32) {
          v[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xv) + (int64_t)r * ldx + c);
        } else {
          const half4_t hv = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(xv) + (int64_t)r * ldx + c);
          v[i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        }
      }
    }
  };
  f32x4 v[ITERS], vn[ITERS];
  load(row, v);
  while (row < rows) {
    const int nrow = row + nw;
    if (nrow < rows) load(nrow, vn);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i)
      if ((i * 64 + lane) * 4 < D) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i)
      if ((i * 64 + lane) * 4 < D) {
        const f32x4 dlt = v[i] - mean;
        q += (dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3]);
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        const f32x4 o = (v[i] - mean) * rstd * g[i] + b[i];
        if (OUT_DT == LMX_F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(yv) + (int64_t)row * ldy + c) = o;
        } else {
          const half4_t h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
          *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(yv) + (int64_t)row * ldy + c) = h;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) v[i] = vn[i];
    row = nrow;
  }
}


__global__ __launch_bounds__(256) void assemble_tokens_kernel(const half_t* __restrict__ patch,
                                                              const float* __restrict__ prefix,
                                                              const float* __restrict__ pos, float* __restrict__ out,
                                                              int B, int np, int n_prefix, int D) {
  const int T = np + n_prefix;
  const int64_t total = (int64_t)B * T * (D / 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (D / 4)) * 4;
    const int64_t bt = i / (D / 4);
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    f32x4 v;
    if (t < n_prefix) {
      v = *reinterpret_cast<const f32x4*>(prefix + (int64_t)t * D + c);
    } else {
      const half4_t h = *reinterpret_cast<const half4_t*>(patch + ((int64_t)b * np + (t - n_prefix)) * D + c);
      v = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    if (pos) v += *reinterpret_cast<const f32x4*>(pos + (int64_t)t * D + c);
    *reinterpret_cast<f32x4*>(out + bt * D + c) = v;
  }
}

// mean over T tokens; block = (b, 256-channel slab of 4-wide lanes): thread owns 4 channels, loops over tokens.
template <int IN_DT>
__global__ __launch_bounds__(64) void token_mean_kernel(const void* __restrict__ xv, float* __restrict__ out, int B,
                                                        int T, int D) {
  const int b = blockIdx.y;
  const int c = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (c >= D) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const int64_t off = ((int64_t)b * T + t) * D + c;
    if (IN_DT == LMX_F32) {
      acc += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xv) + off);
    } else {
      const half4_t h = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(xv) + off);
      acc += f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
  }
  acc /= (float)T;
  *reinterpret_cast<f32x4*>(out + (int64_t)b * D + c) = acc;
}

// DINOv3 RoPE (rotate_half form) on patch tokens, in place.  Thread owns the pair (d, d + hd/2) for 4 d's.
__global__ __launch_bounds__(256) void rope_kernel(half_t* __restrict__ x, int64_t ld, int B, int T, int H, int hd,
                                                   int n_prefix, const float* __restrict__ cos_t,
                                                   const float* __restrict__ sin_t) {
  const int half = hd / 2;
  const int per_tok = H * (half / 4);
  const int np = T - n_prefix;
  const int64_t total = (int64_t)B * np * per_tok;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % per_tok);
    const int64_t bp = i / per_tok;
    const int pidx = (int)(bp % np);
    const int b = (int)(bp / np);
    const int h = w / (half / 4);
    const int d = (w - h * (half / 4)) * 4;
    half_t* base = x + ((int64_t)b * T + n_prefix + pidx) * ld + (int64_t)h * hd;
    const half4_t lo = *reinterpret_cast<const half4_t*>(base + d);
    const half4_t hi = *reinterpret_cast<const half4_t*>(base + d + half);
    const f32x4 c_lo = *reinterpret_cast<const f32x4*>(cos_t + (int64_t)pidx * hd + d);
    const f32x4 c_hi = *reinterpret_cast<const f32x4*>(cos_t + (int64_t)pidx * hd + d + half);
    const f32x4 s_lo = *reinterpret_cast<const f32x4*>(sin_t + (int64_t)pidx * hd + d);
    const f32x4 s_hi = *reinterpret_cast<const f32x4*>(sin_t + (int64_t)pidx * hd + d + half);
    half4_t olo, ohi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = (float)lo[e], bb = (float)hi[e];
      // rotate_half(x) = cat(-x2, x1)
      olo[e] = (half_t)(a * c_lo[e] - bb * s_lo[e]);
      ohi[e] = (half_t)(bb * c_hi[e] + a * s_hi[e]);
    }
    *reinterpret_cast<half4_t*>(base + d) = olo;
    *reinterpret_cast<half4_t*>(base + d + half) = ohi;
  }
}

inline int grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int lmx_k_layernorm(const void* x, int in_dtype, int64_t ldx, const float* gamma, const float* beta,
                               void* y, int out_dtype, int64_t ldy, int rows, int D, float eps, int act, lmx_stream_t stream) {
  LMX_REQUIRE(x && y && gamma && beta, "lmx_k_layernorm: null pointer");
  LMX_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4096, "lmx_k_layernorm: rows=%d D=%d (need D%%4==0, D<=4096)", rows, D);
  LMX_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= D && ldy >= D, "lmx_k_layernorm: strides");
  LMX_REQUIRE(act == LMX_ACT_NONE || act == LMX_ACT_GELU, "lmx_k_layernorm: act %d", act);
  LMX_REQUIRE(aligned16(gamma) && aligned16(beta), "lmx_k_layernorm: gamma/beta alignment");
  LMX_REQUIRE((((uintptr_t)x) & (in_dtype == LMX_F32 ? 15 : 7)) == 0, "lmx_k_layernorm: x alignment");
  LMX_REQUIRE((((uintptr_t)y) & (out_dtype == LMX_F32 ? 15 : 7)) == 0, "lmx_k_layernorm: y alignment");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (D <= 128) {
    dim3 g8((rows + 7) / 8), b256(256);
    if (in_dtype == LMX_F32 && out_dtype == LMX_F16)
      hipLaunchKernelGGL((layernorm_narrow_kernel<LMX_F32, LMX_F16>), g8, b256, 0, st, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
    else if (in_dtype == LMX_F32 && out_dtype == LMX_F32)
      hipLaunchKernelGGL((layernorm_narrow_kernel<LMX_F32, LMX_F32>), g8, b256, 0, st, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
    else if (in_dtype == LMX_F16 && out_dtype == LMX_F16)
      hipLaunchKernelGGL((layernorm_narrow_kernel<LMX_F16, LMX_F16>), g8, b256, 0, st, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
    else if (in_dtype == LMX_F16 && out_dtype == LMX_F32)
      hipLaunchKernelGGL((layernorm_narrow_kernel<LMX_F16, LMX_F32>), g8, b256, 0, st, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
    else
      LMX_REQUIRE(false, "lmx_k_layernorm: bad dtypes %d -> %d", in_dtype, out_dtype);
    return lmx_launch_check("layernorm_narrow_kernel");
  }
  dim3 grid((rows + 3) / 4), block(256);
  // many rows of a ViT width, f32 stream -> f16: the several-rows-per-wave form (LMX_LN_ONE_ROW=1: the one-row-per-wave kernel)
  static int one_row = -1;
  if (one_row < 0) one_row = getenv("LMX_LN_ONE_ROW") ? 1 : 0;
  if (!one_row && rows >= 16384 && act == LMX_ACT_NONE && in_dtype == LMX_F32 && out_dtype == LMX_F16 && D <= 1024) {
    dim3 g2(256 * 8);  // eight workgroups of four waves per CU, each wave walking rows / 8192 rows
    if (D <= 512)
      hipLaunchKernelGGL((layernorm_rows_kernel<LMX_F32, LMX_F16, 2>), g2, block, 0, st, x, ldx, gamma, beta, y, ldy, rows, D, eps);
    else
      hipLaunchKernelGGL((layernorm_rows_kernel<LMX_F32, LMX_F16, 4>), g2, block, 0, st, x, ldx, gamma, beta, y, ldy, rows, D, eps);
    return lmx_launch_check("layernorm_rows_kernel");
  }
#define LMX_LN(IN, OUT, IT) \
  hipLaunchKernelGGL((layernorm_kernel<IN, OUT, IT>), grid, block, 0, st, x, ldx, gamma, beta, y, ldy, rows, D, eps, act)
#define LMX_LN_IT(IN, OUT)        \
  do {                            \
    if (D <= 512)                 \
      LMX_LN(IN, OUT, 2);         \
    else if (D <= 1024)           \
      LMX_LN(IN, OUT, 4);         \
    else                          \
      LMX_LN(IN, OUT, 16);        \
  } while (0)
  if (in_dtype == LMX_F32 && out_dtype == LMX_F16)
    LMX_LN_IT(LMX_F32, LMX_F16);
  else if (in_dtype == LMX_F32 && out_dtype == LMX_F32)
    LMX_LN_IT(LMX_F32, LMX_F32);
  else if (in_dtype == LMX_F16 && out_dtype == LMX_F16)
    LMX_LN_IT(LMX_F16, LMX_F16);
  else if (in_dtype == LMX_F16 && out_dtype == LMX_F32)
    LMX_LN_IT(LMX_F16, LMX_F32);
  else
    LMX_REQUIRE(false, "lmx_k_layernorm: bad dtypes %d -> %d", in_dtype, out_dtype);
#undef LMX_LN_IT
#undef LMX_LN
  return lmx_launch_check("layernorm_kernel");
}

extern "C" int lmx_k_assemble_tokens(const void* patch_f16, const float* prefix, const float* pos, float* out, int B,
                                     int np, int n_prefix, int D, lmx_stream_t stream) {
  LMX_REQUIRE(patch_f16 && out && (prefix || n_prefix == 0), "lmx_k_assemble_tokens: null pointer");
  LMX_REQUIRE(B > 0 && np > 0 && n_prefix >= 0 && D > 0 && D % 4 == 0, "lmx_k_assemble_tokens: shape");
  LMX_REQUIRE(aligned16(out) && ((((uintptr_t)patch_f16) & 7) == 0), "lmx_k_assemble_tokens: alignment");
  const int64_t total = (int64_t)B * (np + n_prefix) * (D / 4);
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const half_t*>(patch_f16), prefix, pos, out, B, np, n_prefix, D);
  return lmx_launch_check("assemble_tokens_kernel");
}

extern "C" int lmx_k_token_mean(const void* x, int in_dtype, float* out, int B, int T, int D, lmx_stream_t stream) {
  LMX_REQUIRE(x && out, "lmx_k_token_mean: null pointer");
  LMX_REQUIRE(B > 0 && T > 0 && D > 0 && D % 4 == 0, "lmx_k_token_mean: shape");
  LMX_REQUIRE(aligned16(out), "lmx_k_token_mean: alignment");
  dim3 grid((D / 4 + 63) / 64, B), block(64);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (in_dtype == LMX_F32)
    hipLaunchKernelGGL((token_mean_kernel<LMX_F32>), grid, block, 0, st, x, out, B, T, D);
  else if (in_dtype == LMX_F16)
    hipLaunchKernelGGL((token_mean_kernel<LMX_F16>), grid, block, 0, st, x, out, B, T, D);
  else
    LMX_REQUIRE(false, "lmx_k_token_mean: bad dtype %d", in_dtype);
  return lmx_launch_check("token_mean_kernel");
}

extern "C" int lmx_k_rope(void* x, int64_t ld, int B, int T, int H, int hd, int n_prefix, const float* cos_t,
                          const float* sin_t, lmx_stream_t stream) {
  LMX_REQUIRE(x && cos_t && sin_t, "lmx_k_rope: null pointer");
  LMX_REQUIRE(B > 0 && T > n_prefix && n_prefix >= 0 && H > 0 && hd % 8 == 0, "lmx_k_rope: shape");
  LMX_REQUIRE(ld % 4 == 0 && ((((uintptr_t)x) & 7) == 0) && aligned16(cos_t) && aligned16(sin_t), "lmx_k_rope: alignment");
  const int64_t total = (int64_t)B * (T - n_prefix) * H * (hd / 8);
  hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<half_t*>(x), ld, B, T, H, hd, n_prefix, cos_t, sin_t);
  return lmx_launch_check("rope_kernel");
}
