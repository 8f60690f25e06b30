// gemm.hip — f16 x f16 -> f32-accumulate GEMM on MFMA (v_mfma_f32_16x16x32_f16), with the epilogues the three
// networks need, and an implicit-GEMM A-operand generator for NHWC 3x3 convolutions.
//
//   C[m][n] = res[m][n] + scale[n] * act( sum_k A[m][k] * W[n][k] + bias[n] )
//
// Replaces (see include/lmx.h): Conv2d+BN+SiLU of YOLOv8, the qkv/proj/fc1/fc2 Linears of
// the SAM / Hiera / DINO ViT blocks (Appendix A.2-A.5).
//
// Design (gfx950):
//   * block tile BM x BN x 64, 256 threads = 4 waves as 2(m) x 2(n); each wave owns (BM/2) x (BN/2) as 16x16 MFMA
//     fragments.  The MFMA is issued "swapped" (a = W fragment, b = A fragment) so that the accumulator holds
//     D[n][m]: a lane then owns 4 CONSECUTIVE output channels of one row -> 8/16-byte epilogue stores and float4
//     bias/scale loads instead of 2-byte scatter.
//   * both operands are K-contiguous, so both LDS images are [rows][64 halfs] (128-B rows) read with
//     ds_read_b128; the 16-B chunk index is XOR-swizzled with (row & 7) which makes every 16-lane ds_read_b128
//     group hit 16 distinct 16-B slots of the 256-B bank row (conflict-free; cdna guide §5.5 T2).
//   * staging is global_load_dwordx4 -> registers -> ds_write_b128 (register staging, not LDS-DMA) because the
//     conv generator needs per-chunk predication (zero padding) and the swizzle is applied on the LDS write;
//     the loads of k-tile t+1 are issued before the MFMAs of tile t and written after them (double-buffered LDS,
//     one barrier per k-tile).
//   * blockIdx -> tile map is XCD-aware (bijective chunking, guide T1): the blocks that land on one XCD walk
//     consecutive n-tiles of the same m-panel so the A panel and W stay in that XCD's L2.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 64;

template <int BM, int BN, int AMODE, int OUT_DT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const lmx_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* As = reinterpret_cast<half_t*>(smem);      // [2][BM][64]
  half_t* Ws = As + 2 * BM * BK;                      // [2][BN][64]
  constexpr int FM = BM / 32;  // 16-row fragments per wave along m
  constexpr int FN = BN / 32;
  constexpr int AI = BM / 32;  // staging passes (32 rows per pass)
  constexpr int WI = BN / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware bijective remap of the block id
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int NT = (p.N + BN - 1) / BN;
  const int mt = swz / NT, nt = swz - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;

  const int c = tid & 7;    // 16-B chunk inside the 64-wide k-tile
  const int r0 = tid >> 3;  // 0..31

  const half_t* A = reinterpret_cast<const half_t*>(p.A);
  const half_t* W = reinterpret_cast<const half_t*>(p.W);

  // per-row state of the A generator
  int64_t a_off[AI];
  int a_iy[AI], a_ix[AI];
  bool a_ok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + r0 + 32 * i;
    a_ok[i] = m < p.M;
    if (AMODE == 0) {
      a_off[i] = (int64_t)m * p.lda;
      a_iy[i] = a_ix[i] = 0;
    } else {
      const int hw = p.Ho * p.Wo;
      const int img = m / hw;
      const int rem = m - img * hw;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      a_off[i] = (int64_t)img * p.H * p.W_ * p.lda;
      a_iy[i] = oy * p.conv_stride - 1;
      a_ix[i] = ox * p.conv_stride - 1;
    }
  }
  int64_t w_off[WI];
  bool w_ok[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int n = n0 + r0 + 32 * i;
    w_ok[i] = n < p.N;
    w_off[i] = (int64_t)n * p.K;
  }

  half8_t a_st[AI], w_st[WI];
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tile = [&](int kt) {
    const int k = kt * BK + c * 8;
    const bool k_ok = k < p.K;
    if (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < AI; ++i)
        a_st[i] = (a_ok[i] && k_ok) ? *reinterpret_cast<const half8_t*>(A + a_off[i] + k) : zero8;
    } else {
      const int tap = k / p.Cin;
      const int ci = k - tap * p.Cin;
      const int ky = tap / 3;
      const int kx = tap - 3 * ky;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
        const bool ok = a_ok[i] && k_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W_;
        a_st[i] = ok ? *reinterpret_cast<const half8_t*>(A + a_off[i] + ((int64_t)iy * p.W_ + ix) * p.lda + ci)
                     : zero8;
      }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
      w_st[i] = (w_ok[i] && k_ok) ? *reinterpret_cast<const half8_t*>(W + w_off[i] + k) : zero8;
  };
  auto store_tile = [&](int buf) {
    half_t* as = As + buf * BM * BK;
    half_t* ws = Ws + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<half8_t*>(as + row * BK + ((c ^ (row & 7)) << 3)) = a_st[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<half8_t*>(ws + row * BK + ((c ^ (row & 7)) << 3)) = w_st[i];
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frow = lane & 15;
  const int fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const half_t* as = As + buf * BM * BK + (wm * (BM / 2) + frow) * BK;
    const half_t* ws = Ws + buf * BN * BK + (wn * (BN / 2) + frow) * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = (((ks << 2) + fq) ^ (frow & 7)) << 3;
      half8_t af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const half8_t*>(as + i * 16 * BK + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const half8_t*>(ws + j * 16 * BK + coff);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane owns row m = ..+(lane&15), channels n..n+3 with n = ..+(lane>>4)*4
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 16 + frow;
    if (m >= p.M) continue;
    const int mr = p.res_rows > 0 ? m % p.res_rows : m;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 16 + fq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        v += b;
      }
      // the same rounding sequence as gemm2_kernel's epilogue (common.h lmx_act; no contraction of scale and residual into
      // one fma; an f16 result is rounded BEFORE its f16 residual is added): the two kernels serve the same shapes at
      // different batch sizes and must agree bit for bit
      if (p.act != LMX_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = lmx_act(v[e], p.act);
      }
      if (p.scale) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(p.scale + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __fmul_rn(v[e], s[e]);
      }
      if (OUT_DT == LMX_F32) {
        float* C = reinterpret_cast<float*>(p.C);
        if (p.res) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (int64_t)mr * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], rr[e]);
        }
        *reinterpret_cast<f32x4*>(C + (int64_t)m * p.ldc + n) = v;
      } else {
        half_t* C = reinterpret_cast<half_t*>(p.C);
        if (p.res) {
          const half4_t rr =
              *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.res) + (int64_t)mr * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __fadd_rn((float)(half_t)v[e], (float)rr[e]);
        }
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        *reinterpret_cast<half4_t*>(C + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
}

template <int BM, int BN, int AMODE, int OUT_DT>
int launch(const lmx_gemm_desc& d, hipStream_t st) {
  const int MT = (d.M + BM - 1) / BM, NT = (d.N + BN - 1) / BN;
  const size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(half_t);
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, AMODE, OUT_DT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, AMODE, OUT_DT>), dim3(MT * NT), dim3(256), smem, st, d);
  return lmx_launch_check("gemm_kernel");
}

template <int AMODE, int OUT_DT>
int dispatch_tile(const lmx_gemm_desc& d, hipStream_t st) {
  // narrow outputs (Detect head 64/80 channels, small stems) take the 128x64 tile
  if (d.N <= 64 || (d.N < 128 && d.N % 128 != 0)) return launch<128, 64, AMODE, OUT_DT>(d, st);
  return launch<128, 128, AMODE, OUT_DT>(d, st);
}

}  // namespace

int lmx_gemm2_launch(const lmx_gemm_desc& d, hipStream_t st);  // gemm2.hip

static bool force_v1() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LMX_GEMM_V1");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

extern "C" int lmx_k_gemm(const lmx_gemm_desc* dp, lmx_stream_t stream) {
  LMX_REQUIRE(dp != nullptr, "lmx_k_gemm: null descriptor");
  const lmx_gemm_desc& d = *dp;
  LMX_REQUIRE(d.A && d.W && d.C, "lmx_k_gemm: null A/W/C");
  LMX_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "lmx_k_gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
  LMX_REQUIRE(d.K % 8 == 0, "lmx_k_gemm: K=%d must be a multiple of 8", d.K);
  LMX_REQUIRE(d.N % 4 == 0, "lmx_k_gemm: N=%d must be a multiple of 4", d.N);
  LMX_REQUIRE(d.lda % 8 == 0, "lmx_k_gemm: lda=%lld must be a multiple of 8", (long long)d.lda);
  LMX_REQUIRE(d.ldc % 4 == 0 && d.ldc >= d.N, "lmx_k_gemm: bad ldc=%lld", (long long)d.ldc);
  LMX_REQUIRE(aligned16(d.A) && aligned16(d.W), "lmx_k_gemm: A/W must be 16-byte aligned");
  LMX_REQUIRE((((uintptr_t)d.C) & 7) == 0, "lmx_k_gemm: C must be 8-byte aligned");
  LMX_REQUIRE(d.out_dtype == LMX_F16 || d.out_dtype == LMX_F32, "lmx_k_gemm: bad out_dtype %d", d.out_dtype);
  LMX_REQUIRE(d.out_dtype == LMX_F16 || aligned16(d.C), "lmx_k_gemm: f32 C must be 16-byte aligned");
  LMX_REQUIRE(d.act >= LMX_ACT_NONE && d.act <= LMX_ACT_RELU, "lmx_k_gemm: bad act %d", d.act);
  if (d.bias) LMX_REQUIRE(aligned16(d.bias), "lmx_k_gemm: bias must be 16-byte aligned");
  if (d.scale) LMX_REQUIRE(aligned16(d.scale), "lmx_k_gemm: scale must be 16-byte aligned");
  if (d.res) LMX_REQUIRE(d.ldr % 4 == 0 && d.ldr >= d.N, "lmx_k_gemm: bad ldr=%lld", (long long)d.ldr);
  LMX_REQUIRE(d.res_rows >= 0, "lmx_k_gemm: res_rows");
  if (d.a_mode == 0) {
    LMX_REQUIRE(d.a_rep > 1 || d.lda >= d.K, "lmx_k_gemm: lda=%lld < K=%d", (long long)d.lda, d.K);
  } else if (d.a_mode == 1) {
    LMX_REQUIRE(d.Cin > 0 && d.Cin % 8 == 0, "lmx_k_gemm: conv Cin=%d must be a multiple of 8", d.Cin);
    LMX_REQUIRE(d.K == 9 * d.Cin, "lmx_k_gemm: conv K=%d != 9*Cin=%d", d.K, 9 * d.Cin);
    LMX_REQUIRE(d.conv_stride == 1 || d.conv_stride == 2, "lmx_k_gemm: conv stride %d", d.conv_stride);
    LMX_REQUIRE(d.H > 0 && d.W_ > 0, "lmx_k_gemm: conv H/W");
    LMX_REQUIRE(d.Ho == (d.H + 2 - 3) / d.conv_stride + 1 && d.Wo == (d.W_ + 2 - 3) / d.conv_stride + 1,
                "lmx_k_gemm: conv Ho/Wo (%d,%d) inconsistent with H/W (%d,%d) stride %d", d.Ho, d.Wo, d.H, d.W_,
                d.conv_stride);
    LMX_REQUIRE(d.M % (d.Ho * d.Wo) == 0, "lmx_k_gemm: conv M=%d not a multiple of Ho*Wo", d.M);
    LMX_REQUIRE(d.lda >= d.Cin, "lmx_k_gemm: conv pixel stride lda=%lld < Cin=%d", (long long)d.lda, d.Cin);
  } else if (d.a_mode == 2) {
    LMX_REQUIRE(d.lda >= d.K, "lmx_k_gemm: lda=%lld < K=%d", (long long)d.lda, d.K);
    LMX_REQUIRE(d.H > 0 && d.W_ > 0 && d.H % 2 == 0 && d.W_ % 2 == 0 && d.M % (d.H * d.W_) == 0,
                "lmx_k_gemm: pooled rows need an even H x W token grid that divides M (H=%d W=%d M=%d)", d.H, d.W_, d.M);
    LMX_REQUIRE(!d.res && !d.scale && d.act == LMX_ACT_NONE, "lmx_k_gemm: pooled rows: no residual / scale / activation");
    LMX_REQUIRE(((int64_t)d.M * d.lda + d.K) * 2 < 0x7fffffffll, "lmx_k_gemm: pooled rows: A must be smaller than 2 GB");
    LMX_REQUIRE(d.M >= 512 && d.N >= 96 && d.N % 8 == 0 && d.ldc % 8 == 0 && aligned16(d.C),
                "lmx_k_gemm: pooled rows are built for the LDS-DMA kernel (M >= 512, N >= 96, N %% 8 == 0): use GEMM + maxpool2 for M=%d N=%d", d.M, d.N);
  } else {
    LMX_REQUIRE(false, "lmx_k_gemm: bad a_mode %d", d.a_mode);
  }
  LMX_REQUIRE(d.a_rep >= 0 && d.a_rep <= 3, "lmx_k_gemm: a_rep=%d (0..3)", d.a_rep);
  LMX_REQUIRE(d.split_k >= 0 && d.split_k <= 64, "lmx_k_gemm: split_k=%d (0..64)", d.split_k);
  if (d.split_k > 1) {
    LMX_REQUIRE(d.out_dtype == LMX_F32 && d.act == LMX_ACT_NONE && d.a_mode != 2 && d.a_rep <= 1, "lmx_k_gemm: split_k needs f32 output, no activation, a_mode 0 or 1");
    LMX_REQUIRE(d.split_stride >= (int64_t)(d.M - 1) * d.ldc + d.N && d.split_stride % 4 == 0, "lmx_k_gemm: split_stride=%lld", (long long)d.split_stride);
    LMX_REQUIRE((d.K + 63) / 64 >= d.split_k, "lmx_k_gemm: split_k=%d exceeds the k-tiles of K=%d", d.split_k, d.K);
    const bool conv_ok2 = d.a_mode == 1 && d.Cin % 32 == 0 && d.H < 32768 && d.W_ < 32768;
    // (any M: whether a layer is split must not depend on the batch, or a frame's bits would; the LDS-DMA kernel zero-fills short tiles)
    LMX_REQUIRE((d.a_mode == 0 || conv_ok2) && d.N >= 64 && d.N % 8 == 0 && d.ldc % 8 == 0 && (!d.res || d.ldr % 8 == 0) &&
                    aligned16(d.C) && (!d.res || aligned16(d.res)),
                "lmx_k_gemm: split_k is built into the LDS-DMA kernel only (N >= 64, N %% 8 == 0; conv: Cin %% 32 == 0); M=%d N=%d", d.M, d.N);
    return lmx_gemm2_launch(d, reinterpret_cast<hipStream_t>(stream));
  }
  if (d.a_rep > 1) {
    LMX_REQUIRE(d.a_mode == 0 && d.K % d.a_rep == 0 && (d.K / d.a_rep) % 64 == 0, "lmx_k_gemm: a_rep=%d needs a_mode 0 and K / a_rep a multiple of 64 (K=%d)", d.a_rep, d.K);
    LMX_REQUIRE(d.lda >= d.K / d.a_rep, "lmx_k_gemm: lda=%lld < K / a_rep", (long long)d.lda);
    LMX_REQUIRE(d.M >= 512 && d.N >= 96 && d.N % 8 == 0 && d.ldc % 8 == 0 && (!d.res || d.ldr % 8 == 0) && aligned16(d.C) && (!d.res || aligned16(d.res)),
                "lmx_k_gemm: a_rep is built into the LDS-DMA kernel only (M >= 512, N >= 96, N %% 8 == 0); M=%d N=%d", d.M, d.N);
    return lmx_gemm2_launch(d, reinterpret_cast<hipStream_t>(stream));
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // large dense problems take the LDS-DMA 256x128 kernel (gemm2.hip); small / narrow ones and the conv generator stay here
  const bool conv_ok = d.a_mode == 1 && d.Cin % 32 == 0 && d.H < 32768 && d.W_ < 32768;  // (f32 out: the exact plan's pre-activations)
  if (d.a_mode == 2) return lmx_gemm2_launch(d, st);
  // (3 x 3 convolutions with 64 output channels — YOLOv8-l's first C2f stage and Detect's box branch: 2.3 M rows at 150 frames — also
  // take the LDS-DMA kernel: half of its 128-wide n-tile is zero-filled, but these launches are bound by A staging, not by the MFMA:
  // LMX_GEMM_N64=0 restores the register-staged kernel for them)
  static int n64 = -1;
  if (n64 < 0) n64 = (getenv("LMX_GEMM_N64") && getenv("LMX_GEMM_N64")[0] == '0') ? 0 : 1;
  const int n_min = (conv_ok && n64 && d.M >= 65536) ? 64 : 96;
  if ((d.a_mode == 0 || conv_ok) && d.M >= 512 && d.N >= n_min && d.N % 8 == 0 && d.ldc % 8 == 0 && (!d.res || d.ldr % 8 == 0) &&
      aligned16(d.C) && (!d.res || aligned16(d.res)) && !force_v1())
    return lmx_gemm2_launch(d, st);
  if (d.a_mode == 0) {
    return d.out_dtype == LMX_F16 ? dispatch_tile<0, LMX_F16>(d, st) : dispatch_tile<0, LMX_F32>(d, st);
  }
  return d.out_dtype == LMX_F16 ? dispatch_tile<1, LMX_F16>(d, st) : dispatch_tile<1, LMX_F32>(d, st);
}
