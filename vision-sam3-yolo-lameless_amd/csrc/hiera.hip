// hiera.hip — the attention half of a Hiera block with 8 x 8-token windows as ONE kernel (stage 1 of Hiera-B+: D = 112, 2 heads of 56;
// TF:models/sam2/modeling_sam2.py Sam2MultiScaleBlock.forward: `hidden_states = residual + proj(attn(qkv(layer_norm1(x))))` with
// window_partition / window_unpartition around the attention, :412-455).
//
// Why: at D = 112 the four launches it replaces — LayerNorm (f32 rows in, f16 rows out), qkv GEMM (f16 rows in, 3D-wide f16 rows
// out), window attention (3D in, D out), projection GEMM (+ f32 residual) — are each bound by HBM and move 672 + 896 + 896 + 1120
// bytes per token between them, of which only the f32 residual stream in and out (448 + 448) is algorithmic.
//
// How: a WAVE owns a window.  It reads the window's 64 f32 rows once — they are both the LayerNorm's input and the residual —,
// normalises them in registers (a token's 112 features sit in four lanes; or takes the f16 rows the previous block's fused MLP
// left), and the f16-rounded result is the register-resident operand of every product; all weights (q | k | v | proj: 124 KB as
// f16 with each head padded to 64 rows) sit in LDS for the life of the workgroup, and nothing a wave computes is ever seen by
// another wave: no barrier, no exchange buffer, no LDS write after the prologue.
//     q^T[d][t] = Wq . X^T     k^T[d][t] = Wk . X^T      (A = weight rows from LDS, B = the window's rows)
//     v[t][d]   = X . Wv^T                               (A = the window's rows — the same registers —, B = weight rows from LDS)
//     S^T[key][query] = K . Q^T                          (A and B are the f16-rounded accumulators of k^T and q^T)
//     P = exp2(S c - max c) in f16; column 63 of v is 1 (bias 1 on a zero weight row), so O^T row 63 is the softmax sum
//     O^T[d][query] = V^T . P^T                          (A = rounded accumulators of v, B = P straight from S's accumulators)
//     x^T[o][t] += Wo[:, head] . O^T / sum               (A = weight rows from LDS, B = rounded accumulators of O^T)
// The accumulator layout of a 16 x 16 tile (lane (c, g) holds rows 4g..4g+3 of column c) is an operand layout of the next MFMA with
// the contraction index permuted (k-slot 8g + 4h + i <-> row 16(2s + h) + 4g + i of k-step s; cdna guide section 3 "An accumulator
// tile as the next MFMA's operand"): both operands of S and of PV come from accumulators and carry the same permutation; for the
// projection the host stores Wo's columns in that order, and — because the f32 rows are read in the accumulator layout of the
// projection they will be added to — the columns of Wq, Wk, Wv as well (lmx/sam.py pack_hiera_attn).
// Rounding points are those of the unfused chain: LayerNorm output, q, k, v, P and the normalised attention output are rounded to
// f16, every sum is f32; the LayerNorm has the arithmetic form of norm.hip (mean, centred sum of squares, 1 / sqrtf(var + eps)).
// (Tried on top and slower: weight fragments through a three-deep register ring with sched_barriers, 531 -> 699 us — hipcc's own
// order, the next read issued behind four MFMAs that are still executing, hides most of the LDS latency —, and requesting the next
// window's rows early, 799 us with spills.)
// One wave per SIMD with the whole 512-register file (X 64, the projection's accumulators 112, P 32, O 64, short-lived tiles 16 - 32).
#include "common.h"

namespace {

constexpr int D = 112, HEADS = 2, HP = 64, KS = 4;
constexpr int ROWB = 256;                     // bytes per LDS weight row: 128 halfs, 16 chunks of 16 B, chunk c of row r at c ^ (r & 15)
constexpr int NQKV = 3 * HEADS * HP;          // 384 rows: q h0 | q h1 | k h0 | k h1 | v h0 | v h1, 64 each (56 real)
constexpr int W_BYTES = NQKV * ROWB;          // 96 KB
constexpr int WO_BYTES = D * ROWB;            // 28 KB: [112 outputs][2 heads x 64 inputs, permuted]
constexpr int B_OFF = W_BYTES + WO_BYTES;
constexpr int SMEM = B_OFF + (NQKV + 3 * D) * 4;  // + biases, LayerNorm gamma and beta (f32): 129 856 bytes

__device__ __forceinline__ half8_t pack8(const f32x4 a, const f32x4 b) {
  return half8_t{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
}
// max(a, b, c) of finite values (or -inf) as two v_med3_f32 against +inf.  NOT inline asm (v_max3_f32): these run on MFMA results, and the
// wait states gfx950 needs between an MFMA's write and a VALU read of the register are inserted by hipcc only in front of
// instructions it knows — an asm v_max3 behind the S MFMAs read accumulators that were not written yet whenever a second wave kept
// the matrix pipe busy (the maximum came out different from run to run: a valid softmax shift, so only the bit-reproducibility test
// saw it, tools/hiera_pool_determinism.py); and not fmaxf, which costs a canonicalising v_max per operand on MFMA results.
__device__ __forceinline__ float hmax3(float a, float b, float c) {
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(a, b, INFINITY), c, INFINITY);
}
// maximum over the four 16-lane rows of a wave, in every row (scalar temporaries: a bit_cast of a vector element reads element 0)
__device__ __forceinline__ float hrow_max4(float ma, float mc) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned u = __builtin_bit_cast(unsigned, fmaxf(ma, mc));
  const u32x2 a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned a0 = a[0], a1 = a[1];
  const unsigned u1 = __builtin_bit_cast(unsigned, fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1)));
  const u32x2 b = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
  const unsigned b0 = b[0], b1 = b[1];
  return fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
}

__device__ __forceinline__ float hrow_sum4(float v) {  // sum over the four 16-lane rows of a wave, in every row
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const u32x2 a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned a0 = a[0], a1 = a[1];
  const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1));
  const u32x2 b = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
  const unsigned b0 = b[0], b1 = b[1];
  return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
}

// INLN: layer_norm1 runs in here (h unused); otherwise h holds its f16 rows (written by the previous block's fused MLP, csrc/mlp.hip)
template <bool INLN>
__global__ __launch_bounds__(256, 1) void hiera_attn8_kernel(const half_t* __restrict__ h, float* __restrict__ x, const int64_t ldx,
                                                              const float* __restrict__ gam, const float* __restrict__ bet, const float eps,
                                                              const half_t* __restrict__ wqkv, const float* __restrict__ bqkv,
                                                              const half_t* __restrict__ wo, const float* __restrict__ bo, const int Gh,
                                                              const int Gw, const int nwin, const float sl2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- prologue: all weights into LDS, swizzled
  for (int i = tid; i < (NQKV + D) * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    const u32x4 v = *reinterpret_cast<const u32x4*>(r < NQKV ? reinterpret_cast<const char*>(wqkv) + (int64_t)i * 16
                                                             : reinterpret_cast<const char*>(wo) + (int64_t)(i - NQKV * 16) * 16);
    *reinterpret_cast<u32x4*>(smem + r * ROWB + ((c ^ (r & 15)) << 4)) = v;
  }
  float* bias = reinterpret_cast<float*>(smem + B_OFF);
  for (int i = tid; i < NQKV + (INLN ? 3 : 1) * D; i += 256)
    bias[i] = i < NQKV ? bqkv[i] : (i < NQKV + D ? bo[i - NQKV] : (i < NQKV + 2 * D ? gam[i - NQKV - D] : bet[i - NQKV - 2 * D]));
  __syncthreads();

  // fragment of weight rows row0 .. row0 + 15, k-step ks: this lane's row is row0 + fr (row0 % 16 == 0), chunk 4 ks + fg
  auto wfrag = [&](const int row0, const int ks) {
    return *reinterpret_cast<const half8_t*>(smem + (row0 + fr) * ROWB + ((((ks << 2) + fg) ^ fr) << 4));
  };
  const int nWx = Gw >> 3, nWy = Gh >> 3;

  for (int w = blockIdx.x * 4 + wave; w < nwin; w += gridDim.x * 4) {
    const int img = w / (nWy * nWx), wi = w - img * (nWy * nWx);
    const int wy = wi / nWx, wx = wi - wy * nWx;
    // token t = 16 tb + fr of the window sits at window row 2 tb + (fr >> 3), column fr & 7
    int64_t row[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) row[tb] = ((int64_t)img * Gh + wy * 8 + 2 * tb + (fr >> 3)) * Gw + wx * 8 + (fr & 7);

    // the window's f32 rows in the projection's accumulator layout: lane (token fr, fg) holds features 16 ob + 4 fg .. + 3.  They are
    // the residual (read once; in the h form long before its first use) and, with INLN, the LayerNorm's input
    f32x4 accp[7][4];
#pragma unroll
    for (int ob = 0; ob < 7; ++ob)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) accp[ob][tb] = *reinterpret_cast<const f32x4*>(x + row[tb] * ldx + ob * 16 + fg * 4);
    half8_t xn[4][KS];
    if constexpr (!INLN) {
      // X: the window's LayerNorm rows as MFMA operand fragments (features 32 ks + 8 fg .. + 7 of token fr; features >= 112 are zeros)
      const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int d = ks * 32 + fg * 8;
          const half8_t v = *reinterpret_cast<const half8_t*>(h + row[tb] * D + (d < D ? d : 0));
          xn[tb][ks] = d < D ? v : zero8;
        }
    } else {
      // LayerNorm (layer_norm1) in registers; rounded to f16 it is X: k-slot 8 fg + 4 hb + i of k-step ks is feature
      // 16 (2 ks + hb) + 4 fg + i (the weights' columns are stored in that order for this form); the eighth block does not exist: zeros
      float mean[4], rstd[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        float sm = 0.f;
#pragma unroll
        for (int ob = 0; ob < 7; ++ob) sm += (accp[ob][tb][0] + accp[ob][tb][1]) + (accp[ob][tb][2] + accp[ob][tb][3]);
        mean[tb] = hrow_sum4(sm) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int ob = 0; ob < 7; ++ob) {
          const f32x4 dl = accp[ob][tb] - mean[tb];
          sq += (dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]);
        }
        rstd[tb] = 1.0f / sqrtf(hrow_sum4(sq) / (float)D + eps);
      }
      f32x4 nrm[8][4];
#pragma unroll
      for (int ob = 0; ob < 7; ++ob) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(bias + NQKV + D + ob * 16 + fg * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + NQKV + 2 * D + ob * 16 + fg * 4);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) nrm[ob][tb] = (accp[ob][tb] - mean[tb]) * rstd[tb] * g + b;
      }
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) nrm[7][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xn[tb][ks] = pack8(nrm[2 * ks][tb], nrm[2 * ks + 1][tb]);
    }
    // the projection's accumulators start from the residual + bias
#pragma unroll
    for (int ob = 0; ob < 7; ++ob) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + NQKV + ob * 16 + fg * 4);
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) accp[ob][tb] += bv;
    }
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int hh = 0; hh < HEADS; ++hh) {
      // Live ranges are kept short on purpose (at their natural extent the kernel needed 463 registers and ~850 v_accvgpr moves):
      // q and k two row blocks — one k-step of S — at a time, S and the softmax before v exists, v one d block at a time with its PV
      // products issued at once.
      half8_t pf[4][2];  // P^T[query block][k-step over keys]
      {
        half8_t qf[4][2], kf[4][2];
        // ---- q^T and k^T: [64 d][64 tokens] each; rounded to f16 they are the B (q) and A (k) operands of S
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int r0 = sec * (HEADS * HP) + hh * HP + s * 32;
            f32x4 acc[2][4];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + r0 + r2 * 16 + fg * 4);
#pragma unroll
              for (int tb = 0; tb < 4; ++tb) acc[r2][tb] = bv;
            }
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                const half8_t a = wfrag(r0 + r2 * 16, ks);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) acc[r2][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[tb][ks], acc[r2][tb], 0, 0, 0);
              }
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
              const half8_t f = pack8(acc[0][tb], acc[1][tb]);
              if (sec == 0)
                qf[tb][s] = f;
              else
                kf[tb][s] = f;
            }
          }
        // ---- S^T[key][query] = K . Q^T one query block at a time, softmax over the 64 keys of a query (16 in this lane, 4 lanes per query)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          f32x4 sacc[4];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            sacc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][s], qf[qb][s], sacc[kb], 0, 0, 0);
          }
          float ma = hmax3(sacc[0][0], sacc[0][1], sacc[0][2]);
          float mc = hmax3(sacc[2][0], sacc[2][1], sacc[2][2]);
          ma = hmax3(ma, sacc[0][3], sacc[1][0]);
          mc = hmax3(mc, sacc[2][3], sacc[3][0]);
          ma = hmax3(ma, sacc[1][1], sacc[1][2]);
          mc = hmax3(mc, sacc[3][1], sacc[3][2]);
          ma = hmax3(ma, sacc[1][3], sacc[3][3]);
          const float nmb = -(hrow_max4(ma, mc) * sl2);
          f32x4 e[4];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) e[kb][r] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], sl2, nmb));
#pragma unroll
          for (int s = 0; s < 2; ++s) pf[qb][s] = pack8(e[2 * s], e[2 * s + 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // phase boundary: q and k are dead from here
      // ---- v: [64 tokens][64 d] one d block at a time (tokens are the MFMA rows); rounded, it is the A operand of
      // O^T[d][query] = V^T . P^T; row 63 of O^T is the softmax sum (v's column 63 is the constant 1); normalise, round
      half8_t of[4][2];
      {
        const int r0 = 2 * (HEADS * HP) + hh * HP;
        f32x4 oacc[4][4];  // [d block][query block]
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const float b = bias[r0 + db * 16 + fr];
          f32x4 acc[4];  // [token block]
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[tb] = f32x4{b, b, b, b};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const half8_t bw = wfrag(r0 + db * 16, ks);
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xn[tb][ks], bw, acc[tb], 0, 0, 0);
          }
          const half8_t v0 = pack8(acc[0], acc[1]), v1 = pack8(acc[2], acc[3]);
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) {
            oacc[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, pf[qb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            oacc[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, pf[qb][1], oacc[db][qb], 0, 0, 0);
          }
        }
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          const float l = __shfl(oacc[3][qb][3], 48 + fr, 64);  // O^T[63][query fr]: lane group 3, register 3 of d block 3
          const float inv = 1.0f / l;
#pragma unroll
          for (int s = 0; s < 2; ++s) of[qb][s] = pack8(oacc[2 * s][qb] * inv, oacc[2 * s + 1][qb] * inv);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // phase boundary: nothing moves across (live ranges stay those of the phases)
      // ---- x^T[o][token] += Wo[:, head hh] . O^T   (Wo's columns of a head are stored in the operand's k-slot order; column 63 is zero)
#pragma unroll
      for (int ob = 0; ob < 7; ++ob)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8_t a = wfrag(NQKV + ob * 16, hh * 2 + s);
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) accp[ob][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, of[tb][s], accp[ob][tb], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- store the residual stream: lane (token fr, fg) holds features 16 ob + 4 fg .. + 3 of its four tokens
#pragma unroll
    for (int ob = 0; ob < 7; ++ob)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) *reinterpret_cast<f32x4*>(x + row[tb] * ldx + ob * 16 + fg * 4) = accp[ob][tb];
  }
}

}  // namespace

// x f32 [rows, ldx] updated in place, rows = n_img * Gh * Gw on a token grid whose sides are multiples of 8.  h f16 [rows, 112] =
// layer_norm1(x), or NULL: the kernel then normalises x itself with gamma, beta f32 [112], eps (and wqkv_p's columns are in k-slot
// order).  wqkv_p f16 [384, 128], bqkv_p f32 [384], wo_p f16 [112, 128], bo f32 [112]: lmx/sam.py pack_hiera_attn.
extern "C" int lmx_k_hiera_attn8(const void* h, float* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* wqkv_p,
                                 const float* bqkv_p, const void* wo_p, const float* bo, int n_img, int Gh, int Gw, int D_, int heads,
                                 float scale, lmx_stream_t stream) {
  LMX_REQUIRE(x && (h || (gamma && beta)) && wqkv_p && bqkv_p && wo_p && bo, "lmx_k_hiera_attn8: null pointer");
  LMX_REQUIRE(D_ == D && heads == HEADS, "lmx_k_hiera_attn8: built for D = 112 with 2 heads, got D = %d heads = %d", D_, heads);
  LMX_REQUIRE(n_img > 0 && Gh > 0 && Gw > 0 && Gh % 8 == 0 && Gw % 8 == 0, "lmx_k_hiera_attn8: token grid %d x %d is not whole 8 x 8 windows", Gh, Gw);
  LMX_REQUIRE(ldx >= D && ldx % 4 == 0 && aligned16(x) && aligned16(h) && aligned16(wqkv_p) && aligned16(wo_p), "lmx_k_hiera_attn8: ldx / alignment");
  const int64_t nwin = (int64_t)n_img * (Gh / 8) * (Gw / 8);
  LMX_REQUIRE(nwin < (1ll << 31), "lmx_k_hiera_attn8: too many windows");
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera_attn8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera_attn8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    LMX_HIP(hipGetDevice(&dev));
    LMX_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int64_t need = (nwin + 3) / 4;
  const unsigned grid = (unsigned)(need < n_cu ? need : n_cu);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const half_t* hp = reinterpret_cast<const half_t*>(h);
  const half_t *wq = reinterpret_cast<const half_t*>(wqkv_p), *wop = reinterpret_cast<const half_t*>(wo_p);
  const float sl2 = scale * 1.44269504088896340736f;
  if (h)
    hipLaunchKernelGGL(hiera_attn8_kernel<false>, dim3(grid), dim3(256), SMEM, st, hp, x, ldx, gamma, beta, eps, wq, bqkv_p, wop, bo, Gh, Gw, (int)nwin, sl2);
  else
    hipLaunchKernelGGL(hiera_attn8_kernel<true>, dim3(grid), dim3(256), SMEM, st, hp, x, ldx, gamma, beta, eps, wq, bqkv_p, wop, bo, Gh, Gw, (int)nwin, sl2);
  return lmx_launch_check("hiera_attn8_kernel");
}

// ================================================================================================================================
// Stage 2 of Hiera-B+ (D = 224, 4 heads of 56, 4 x 4-token windows): the same idea — a window's rows stay in registers from the qkv
// projection through attention and the output projection — but the weights (4 heads x (q | k | v | proj) = 16 matrices of 32 KB)
// do not fit in LDS and STREAM through a four-slot ring by LDS-DMA, one matrix per step, exactly as csrc/mlp.hip streams its chunks:
// counted s_waitcnt vmcnt for the wave's own pieces, ONE raw s_barrier per matrix, the slot freed by that barrier refilled with the
// matrix three steps ahead.  The host stores every matrix as its LDS image (lmx/sam.py pack_hiera_attn4), so a DMA piece is a
// linear 1 KB copy.  A workgroup = 8 waves x 2 windows = 256 tokens per pass over the weights (512 KB from L2); a window is one
// 16-token MFMA block, so S and PV are block-diagonal: S^T = K . Q^T is one 16 x 16 tile per (window, head), PV runs on
// v_mfma_f32_16x16x16_f16 whose four k-slots per lane ARE the accumulator layout (keys 4g .. 4g+3) — no padding of the 16 keys.
// layer_norm1's rows come from the previous block's fused MLP (h_next).  Rounding points as in the unfused chain.
namespace {

#ifndef LMX_H4_TB
#define LMX_H4_TB 2
#endif
constexpr int D4 = 224, HEADS4 = 4, KS4 = 7, TB4 = LMX_H4_TB, NW4 = 16 / TB4;
constexpr int MAT = 32768;                 // bytes of one matrix image (q | k | v: 64 rows x 512 B; proj: 256 rows x 128 B)
constexpr int NST4 = 4, LA4 = NST4 - 1, PT4 = MAT / (NW4 * 1024);  // ring slots, matrices in flight, DMA pieces per wave and matrix
constexpr int NB4 = HEADS4 * 3 * 64 + D4;  // biases: [head][q | k | v][64] then the projection's
constexpr int SMEM4 = NST4 * MAT + NB4 * 4;

typedef __fp16 half4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NW4 * 64, NW4 / 4) void hiera_attn4_kernel(const half_t* __restrict__ h, float* __restrict__ x, const int64_t ldx,
                                                                   const half_t* __restrict__ img, const float* __restrict__ bias_g,
                                                                   const int Gh, const int Gw, const int nwin, const int ngroup,
                                                                   const float sl2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias = reinterpret_cast<float*>(smem + NST4 * MAT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  for (int i = tid; i < NB4; i += NW4 * 64) bias[i] = bias_g[i];

  // ---- the weight stream: matrix c (counted over this workgroup's groups) is image c % 16 and lands in slot c % NST4
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(img), 0, 16 * MAT, 0x00020000);
  const int my_groups = (ngroup - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ctotal = my_groups * 16;
  const unsigned voff = (unsigned)(wave * (PT4 * 1024) + lane * 16);
  auto issue = [&](const int c) {
    char* dst = smem + (c % NST4) * MAT + wave * (PT4 * 1024);
#pragma unroll
    for (int t = 0; t < PT4; ++t) lds_dma16(w_rs, dst + t * 1024, voff + t * 1024, (c & 15) * MAT);
  };
#pragma unroll
  for (int c = 0; c < LA4; ++c)
    if (c < ctotal) issue(c);
  __syncthreads();  // the biases are in LDS for every wave (the first reads come before the stream's first barrier)

  const int nWx = Gw >> 2, nWy = Gh >> 2;
  int c = 0;
  for (int grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
    // this wave's two windows; a window past the last one computes on window 0 and stores nothing
    int64_t row[TB4];
    bool live[TB4];
#pragma unroll
    for (int tb = 0; tb < TB4; ++tb) {
      const int w = grp * (NW4 * TB4) + wave * TB4 + tb;
      live[tb] = w < nwin;
      const int wc = live[tb] ? w : 0;
      const int im = wc / (nWy * nWx), wi = wc - im * (nWy * nWx);
      const int wy = wi / nWx, wx = wi - wy * nWx;
      row[tb] = ((int64_t)im * Gh + wy * 4 + (fr >> 2)) * Gw + wx * 4 + (fr & 3);
    }
    half8_t xn[TB4][KS4];
#pragma unroll
    for (int tb = 0; tb < TB4; ++tb)
#pragma unroll
      for (int ks = 0; ks < KS4; ++ks) xn[tb][ks] = *reinterpret_cast<const half8_t*>(h + row[tb] * D4 + ks * 32 + fg * 8);
    f32x4 accp[14][TB4];
#pragma unroll
    for (int ob = 0; ob < 14; ++ob) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + HEADS4 * 192 + ob * 16 + fg * 4);
#pragma unroll
      for (int tb = 0; tb < TB4; ++tb) accp[ob][tb] = *reinterpret_cast<const f32x4*>(x + row[tb] * ldx + ob * 16 + fg * 4) + bv;
    }

    // one step of the stream: matrix c has landed for every wave, every wave is done with matrix c - 1, whose slot is refilled
    auto step = [&]() -> const char* {
      const int left = ctotal - 1 - c;
      wait_tiles<PT4>(left < LA4 - 1 ? left : LA4 - 1);
      __builtin_amdgcn_s_barrier();
      if (c + LA4 < ctotal) issue(c + LA4);
      const char* m = smem + (c % NST4) * MAT;
      ++c;
      return m;
    };

#pragma unroll 1
    for (int hh = 0; hh < HEADS4; ++hh) {
      half8_t qf[TB4][2], kf[TB4][2];
#pragma unroll
      for (int sec = 0; sec < 2; ++sec) {  // q^T, k^T: [64 d][16 tokens] per window
        const char* m = step();
        f32x4 acc[4][TB4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + hh * 192 + sec * 64 + rb * 16 + fg * 4);
#pragma unroll
          for (int tb = 0; tb < TB4; ++tb) acc[rb][tb] = bv;
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int ks = 0; ks < KS4; ++ks) {
            const half8_t a = *reinterpret_cast<const half8_t*>(m + (rb * 16 + fr) * 512 + ((((ks << 2) + fg) ^ fr) << 4));
#pragma unroll
            for (int tb = 0; tb < TB4; ++tb) acc[rb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[tb][ks], acc[rb][tb], 0, 0, 0);
          }
#pragma unroll
        for (int tb = 0; tb < TB4; ++tb)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const half8_t f = pack8(acc[2 * s][tb], acc[2 * s + 1][tb]);
            if (sec == 0)
              qf[tb][s] = f;
            else
              kf[tb][s] = f;
          }
      }
      // S^T[key][query] of each window and its softmax need no weights: done before v's image is consumed, so that q and k are dead
      // by then.  Lane (query fr, fg) holds keys 4 fg .. 4 fg + 3.
      half4v pf[TB4];
#pragma unroll
      for (int tb = 0; tb < TB4; ++tb) {
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[tb][s], qf[tb][s], sc, 0, 0, 0);
        const float nmb = -(hrow_max4(__builtin_amdgcn_fmed3f(sc[0], sc[1], INFINITY), __builtin_amdgcn_fmed3f(sc[2], sc[3], INFINITY)) * sl2);
#pragma unroll
        for (int r = 0; r < 4; ++r) pf[tb][r] = (half_t)__builtin_amdgcn_exp2f(fmaf(sc[r], sl2, nmb));
      }
      half8_t of[TB4][2];
      {  // v: [16 tokens][64 d] per window (tokens are the MFMA rows), one d block at a time, each straight into
         // O^T[d][query] = V^T . P^T over the 16 keys: v_mfma_f32_16x16x16_f16, k-slot 4 g + i = key 4 g + i (both accumulator layouts)
        const char* m = step();
        f32x4 oacc[TB4][4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const float b = bias[hh * 192 + 128 + db * 16 + fr];
          f32x4 acc[TB4];
#pragma unroll
          for (int tb = 0; tb < TB4; ++tb) acc[tb] = f32x4{b, b, b, b};
#pragma unroll
          for (int ks = 0; ks < KS4; ++ks) {
            const half8_t bw = *reinterpret_cast<const half8_t*>(m + (db * 16 + fr) * 512 + ((((ks << 2) + fg) ^ fr) << 4));
#pragma unroll
            for (int tb = 0; tb < TB4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xn[tb][ks], bw, acc[tb], 0, 0, 0);
          }
#pragma unroll
          for (int tb = 0; tb < TB4; ++tb) {
            const half4v vf = {(half_t)acc[tb][0], (half_t)acc[tb][1], (half_t)acc[tb][2], (half_t)acc[tb][3]};
            oacc[tb][db] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf[tb], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          }
        }
#pragma unroll
        for (int tb = 0; tb < TB4; ++tb) {
          const float l = __shfl(oacc[tb][3][3], 48 + fr, 64);  // row 63: the softmax sum (v's column 63 is the constant 1)
          const float inv = 1.0f / l;
#pragma unroll
          for (int s = 0; s < 2; ++s) of[tb][s] = pack8(oacc[tb][2 * s] * inv, oacc[tb][2 * s + 1] * inv);
        }
      }
      {  // x^T[o][token] += Wo[:, head hh] . O^T  (image rows of 128 B: 8 chunks, chunk c of row r at c ^ ((r >> 1) & 7))
        const char* m = step();
#pragma unroll
        for (int ob = 0; ob < 14; ++ob)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const half8_t a = *reinterpret_cast<const half8_t*>(m + (ob * 16 + fr) * 128 + ((((s << 2) + fg) ^ ((fr >> 1) & 7)) << 4));
#pragma unroll
            for (int tb = 0; tb < TB4; ++tb) accp[ob][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, of[tb][s], accp[ob][tb], 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int tb = 0; tb < TB4; ++tb)
      if (live[tb]) {
#pragma unroll
        for (int ob = 0; ob < 14; ++ob) *reinterpret_cast<f32x4*>(x + row[tb] * ldx + ob * 16 + fg * 4) = accp[ob][tb];
      }
  }
}

}  // namespace

// h f16 [rows, 224] contiguous = layer_norm1(x); x f32 [rows, ldx] updated in place; rows = n_img * Gh * Gw, Gh and Gw multiples of 4.
// w_img f16 [16][16384]: the LDS images of head h's q, k, v and projection matrices at index 4 h + {0, 1, 2, 3}; bias f32
// [4 * 192 + 224] (lmx/sam.py pack_hiera_attn4).
extern "C" int lmx_k_hiera_attn4(const void* h, float* x, int64_t ldx, const void* w_img, const float* bias, int n_img, int Gh, int Gw,
                                 int D_, int heads, float scale, lmx_stream_t stream) {
  LMX_REQUIRE(h && x && w_img && bias, "lmx_k_hiera_attn4: null pointer");
  LMX_REQUIRE(D_ == D4 && heads == HEADS4, "lmx_k_hiera_attn4: built for D = 224 with 4 heads, got D = %d heads = %d", D_, heads);
  LMX_REQUIRE(n_img > 0 && Gh > 0 && Gw > 0 && Gh % 4 == 0 && Gw % 4 == 0, "lmx_k_hiera_attn4: token grid %d x %d is not whole 4 x 4 windows", Gh, Gw);
  LMX_REQUIRE(ldx >= D4 && ldx % 4 == 0 && aligned16(h) && aligned16(x) && aligned16(w_img), "lmx_k_hiera_attn4: ldx / alignment");
  const int64_t nwin = (int64_t)n_img * (Gh / 4) * (Gw / 4);
  LMX_REQUIRE(nwin < (1ll << 31), "lmx_k_hiera_attn4: too many windows");
  const int64_t ngroup = (nwin + NW4 * TB4 - 1) / (NW4 * TB4);
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera_attn4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM4));
    attr_set = true;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    LMX_HIP(hipGetDevice(&dev));
    LMX_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const unsigned grid = (unsigned)(ngroup < n_cu ? ngroup : n_cu);
  hipLaunchKernelGGL(hiera_attn4_kernel, dim3(grid), dim3(NW4 * 64), SMEM4, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const half_t*>(h), x, ldx, reinterpret_cast<const half_t*>(w_img), bias, Gh, Gw, (int)nwin, (int)ngroup,
                     scale * 1.44269504088896340736f);
  return lmx_launch_check("hiera_attn4_kernel");
}

// ================================================================================================================================
// The block that opens stage 2 of Hiera-B+ (112 -> 224 channels, 4 heads of 56): keys and values are the 8 x 8 = 64 tokens of a
// window, queries and the shortcut are their 2 x 2 max-pools (16 per window; TF:models/sam2/modeling_sam2.py Sam2MultiScaleBlock
// .forward: `residual = do_pool(proj(hidden_states))`, Sam2MultiScaleAttention with q_stride: `query = do_pool(query)`), i.e. the five
// launches shortcut GEMM (+ pool), q GEMM (+ pool), k | v GEMM, window attention with pooled queries, projection GEMM (+ residual).
// One kernel, a wave per window, weights streamed as in hiera_attn4_kernel (14 images of 32 KB per pass of 8 windows).
// The pooling costs nothing: the window's tokens are assigned to MFMA columns so that the four partners of a 2 x 2 pool sit in the
// SAME lane of four different token blocks — block tb holds sub-position (tb >> 1, tb & 1) of pooled token fr — so a pool is an
// element-wise maximum over four accumulator tiles and its result is already the 16-query operand / the 16-token output tile.
// (max and the f16 rounding commute, so pooling the f32 accumulators equals pooling the rounded q as the unfused chain does.)
namespace {

#ifndef LMX_HP_NW
#define LMX_HP_NW 8
#endif
constexpr int DI = 112, DO = 224, HEADSP = 4, KSP = 4, NWP = LMX_HP_NW, NIMG = 14;
constexpr int NSTP = 4, LAP = NSTP - 1, PTP = MAT / (NWP * 1024);
constexpr int NBP = DO + HEADSP * 192 + DO;  // biases: shortcut [224], [head][q | k | v][64], projection [224]
constexpr int SMEMP = NSTP * MAT + NBP * 4;

__device__ __forceinline__ f32x4 vmax4(const f32x4 a, const f32x4 b, const f32x4 c, const f32x4 d) {
  return f32x4{fmaxf(fmaxf(a[0], b[0]), fmaxf(c[0], d[0])), fmaxf(fmaxf(a[1], b[1]), fmaxf(c[1], d[1])), fmaxf(fmaxf(a[2], b[2]), fmaxf(c[2], d[2])),
               fmaxf(fmaxf(a[3], b[3]), fmaxf(c[3], d[3]))};
}

__global__ __launch_bounds__(NWP * 64, NWP / 4) void hiera_attnp_kernel(const half_t* __restrict__ h, float* __restrict__ out,
                                                                         const half_t* __restrict__ img, const float* __restrict__ bias_g,
                                                                         const int Gh, const int Gw, const int nwin, const int ngroup,
                                                                         const float sl2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias = reinterpret_cast<float*>(smem + NSTP * MAT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  for (int i = tid; i < NBP; i += NWP * 64) bias[i] = bias_g[i];

  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(img), 0, NIMG * MAT, 0x00020000);
  const int my_groups = (ngroup - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ctotal = my_groups * NIMG;
  const unsigned voff = (unsigned)(wave * (PTP * 1024) + lane * 16);
  auto issue = [&](const int c) {
    char* dst = smem + (c % NSTP) * MAT + wave * (PTP * 1024);
#pragma unroll
    for (int t = 0; t < PTP; ++t) lds_dma16(w_rs, dst + t * 1024, voff + t * 1024, (c % NIMG) * MAT);
  };
#pragma unroll
  for (int c = 0; c < LAP; ++c)
    if (c < ctotal) issue(c);
  __syncthreads();  // the biases are in LDS for every wave

  const int nWx = Gw >> 3, nWy = Gh >> 3, Go = Gw >> 1;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  int c = 0;
  for (int grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
    const int w = grp * NWP + wave;
    const bool live = w < nwin;
    const int wc = live ? w : 0;
    const int im = wc / (nWy * nWx), wi = wc - im * (nWy * nWx);
    const int wy = wi / nWx, wx = wi - wy * nWx;
    // pooled token fr = (py, px) = (fr >> 2, fr & 3) of the window; block tb holds its partner (2 py + (tb >> 1), 2 px + (tb & 1))
    half8_t xn[4][KSP];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      const int64_t row = ((int64_t)im * Gh + wy * 8 + 2 * (fr >> 2) + (tb >> 1)) * Gw + wx * 8 + 2 * (fr & 3) + (tb & 1);
#pragma unroll
      for (int ks = 0; ks < KSP; ++ks) {
        const int d = ks * 32 + fg * 8;
        const half8_t v = *reinterpret_cast<const half8_t*>(h + row * DI + (d < DI ? d : 0));
        xn[tb][ks] = d < DI ? v : zero8;
      }
    }
    auto step = [&]() -> const char* {
      const int left = ctotal - 1 - c;
      wait_tiles<PTP>(left < LAP - 1 ? left : LAP - 1);
      __builtin_amdgcn_s_barrier();
      if (c + LAP < ctotal) issue(c + LAP);
      const char* m = smem + (c % NSTP) * MAT;
      ++c;
      return m;
    };
    // fragment of image rows row0 .. row0 + 15 (256-byte rows), k-step ks
    auto frag = [&](const char* m, const int row0, const int ks) {
      return *reinterpret_cast<const half8_t*>(m + (row0 + fr) * 256 + ((((ks << 2) + fg) ^ fr) << 4));
    };

    // ---- the shortcut: proj(h) pooled, + its bias + the output projection's bias: the accumulators of the block's output
    f32x4 accp[14];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const char* m = step();
#pragma unroll
      for (int rb = half * 8; rb < (half ? 14 : 8); ++rb) {
        f32x4 acc[4];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) {
          const half8_t a = frag(m, (rb - half * 8) * 16, ks);
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[tb][ks], acc[tb], 0, 0, 0);
        }
        accp[rb] = vmax4(acc[0], acc[1], acc[2], acc[3]) + *reinterpret_cast<const f32x4*>(bias + rb * 16 + fg * 4);  // shortcut + projection bias
      }
    }
#pragma unroll 1
    for (int hh = 0; hh < HEADSP; ++hh) {
      const float* bh = bias + DO + hh * 192;
      half8_t pf[2];  // P^T of the head: the 16 pooled queries against the 64 keys (k-slot order = key order of the v accumulators)
      {  // image [q | k] of the head: q^T pooled (16 queries), k^T (64 keys); two row blocks (one k-step of S) at a time: 32 live
         // accumulator registers instead of 64 (at 64 the kernel spilled, and the scratch traffic reached HBM: 1.3 GB written per
         // launch for 0.44 GB of output, profiles/r03_pmc_summary.txt of the first build)
        half8_t qf[2], kf[4][2];
        const char* m = step();
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            f32x4 acc[2][4];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
              for (int tb = 0; tb < 4; ++tb) acc[r2][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
              for (int ks = 0; ks < KSP; ++ks) {
                const half8_t a = frag(m, sec * 64 + (2 * s + r2) * 16, ks);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) acc[r2][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[tb][ks], acc[r2][tb], 0, 0, 0);
              }
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bh + sec * 64 + (2 * s) * 16 + fg * 4);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bh + sec * 64 + (2 * s + 1) * 16 + fg * 4);
            if (sec == 0) {
              qf[s] = pack8(vmax4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]) + b0, vmax4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]) + b1);
            } else {
#pragma unroll
              for (int tb = 0; tb < 4; ++tb) kf[tb][s] = pack8(acc[0][tb] + b0, acc[1][tb] + b1);
            }
          }
        // S and the softmax need no weights: done here, so that q and k are dead before v's image is consumed
        f32x4 sacc[4];  // S^T[key block][the 16 queries]
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          sacc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 2; ++s) sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][s], qf[s], sacc[kb], 0, 0, 0);
        }
        float ma = hmax3(sacc[0][0], sacc[0][1], sacc[0][2]);
        float mc = hmax3(sacc[2][0], sacc[2][1], sacc[2][2]);
        ma = hmax3(ma, sacc[0][3], sacc[1][0]);
        mc = hmax3(mc, sacc[2][3], sacc[3][0]);
        ma = hmax3(ma, sacc[1][1], sacc[1][2]);
        mc = hmax3(mc, sacc[3][1], sacc[3][2]);
        ma = hmax3(ma, sacc[1][3], sacc[3][3]);
        const float nmb = -(hrow_max4(ma, mc) * sl2);
        f32x4 e[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) e[kb][r] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], sl2, nmb));
#pragma unroll
        for (int s = 0; s < 2; ++s) pf[s] = pack8(e[2 * s], e[2 * s + 1]);
      }
      half8_t of[2];
      {  // image [v | -]: v [64 keys][64 d] one d block at a time, then the head's attention: 16 pooled queries against the window's 64 keys
        const char* m = step();
        f32x4 oacc[4];  // O^T[d block][the 16 queries]
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const float b = bh[128 + db * 16 + fr];
          f32x4 acc[4];  // v[token block][this d block]
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[tb] = f32x4{b, b, b, b};
#pragma unroll
          for (int ks = 0; ks < KSP; ++ks) {
            const half8_t bw = frag(m, db * 16, ks);
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xn[tb][ks], bw, acc[tb], 0, 0, 0);
          }
          oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(acc[0], acc[1]), pf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(acc[2], acc[3]), pf[1], oacc[db], 0, 0, 0);
        }
        const float l = __shfl(oacc[3][3], 48 + fr, 64);
        const float inv = 1.0f / l;
#pragma unroll
        for (int s = 0; s < 2; ++s) of[s] = pack8(oacc[2 * s] * inv, oacc[2 * s + 1] * inv);
      }
      {  // image Wo[:, head]: 128-byte rows
        const char* m = step();
#pragma unroll
        for (int ob = 0; ob < 14; ++ob)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const half8_t a = *reinterpret_cast<const half8_t*>(m + (ob * 16 + fr) * 128 + ((((s << 2) + fg) ^ ((fr >> 1) & 7)) << 4));
            accp[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, of[s], accp[ob], 0, 0, 0);
          }
      }
    }
    if (live) {
      const int64_t orow = ((int64_t)im * (Gh >> 1) + wy * 4 + (fr >> 2)) * Go + wx * 4 + (fr & 3);
#pragma unroll
      for (int ob = 0; ob < 14; ++ob) *reinterpret_cast<f32x4*>(out + orow * DO + ob * 16 + fg * 4) = accp[ob];
    }
  }
}

}  // namespace

// ================================================================================================================================
// The block that opens stage 3 (224 -> 448 channels, 8 heads of 56): the same operation on 4 x 4-token windows — 16 keys / values, 4
// pooled queries and shortcut rows per window.  A wave takes FOUR windows: lane fr is pooled token fr & 3 of window fr >> 2, block tb
// its 2 x 2 partner (tb >> 1, tb & 1), so the pools are again element-wise maxima and the 16 pooled tokens of the four windows form
// one MFMA block.  The four windows share the S and PV tiles: S^T[key][query] holds the products of every key with every query of
// the four windows, and a lane keeps only those of its own window — key rows 4 fg .. 4 fg + 3 of a block belong to window fg,
// query fr to window fr >> 2 — the others are set to -inf before the softmax (P = 0: they drop out of PV and of the row sum).
// 47 weight images per pass (shortcut 7, per head q, k, v and two halves of the projection's columns), 4 waves with the whole
// register file (X alone is 112 registers, the output accumulators another 112; hipcc spills ~150 registers per lane, all of it in
// the per-group prologue / shortcut / epilogue code — none inside the head loop).
namespace {

constexpr int DI2 = 224, DO2 = 448, HEADS2 = 8, KS2 = 7, NW2 = 4, NIMG2 = 7 + 5 * HEADS2;
constexpr int NST2 = 4, LA2 = NST2 - 1, PT2 = MAT / (NW2 * 1024);
constexpr int NB2 = DO2 + HEADS2 * 192;  // biases: shortcut + projection [448], [head][q | k | v][64]
constexpr int SMEM2 = NST2 * MAT + NB2 * 4;

__global__ __launch_bounds__(NW2 * 64, 1) void hiera_attnq_kernel(const half_t* __restrict__ h, float* __restrict__ out,
                                                                   const half_t* __restrict__ img, const float* __restrict__ bias_g,
                                                                   const int Gh, const int Gw, const int nwin, const int ngroup,
                                                                   const float sl2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias = reinterpret_cast<float*>(smem + NST2 * MAT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  for (int i = tid; i < NB2; i += NW2 * 64) bias[i] = bias_g[i];

  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(img), 0, NIMG2 * MAT, 0x00020000);
  const int my_groups = (ngroup - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ctotal = my_groups * NIMG2;
  const unsigned voff = (unsigned)(wave * (PT2 * 1024) + lane * 16);
  auto issue = [&](const int c) {
    char* dst = smem + (c % NST2) * MAT + wave * (PT2 * 1024);
#pragma unroll
    for (int t = 0; t < PT2; ++t) lds_dma16(w_rs, dst + t * 1024, voff + t * 1024, (c % NIMG2) * MAT);
  };
#pragma unroll
  for (int c = 0; c < LA2; ++c)
    if (c < ctotal) issue(c);
  __syncthreads();  // the biases are in LDS for every wave

  const int nWx = Gw >> 2, nWy = Gh >> 2, Go = Gw >> 1;
  const bool mine = fg == (fr >> 2);  // S^T rows of this lane (keys of window fg) against its query's window
  int c = 0;
  for (int grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
    const int w = grp * (NW2 * 4) + wave * 4 + (fr >> 2);
    const bool live = w < nwin;
    const int wc = live ? w : 0;
    const int im = wc / (nWy * nWx), wi = wc - im * (nWy * nWx);
    const int wy = wi / nWx, wx = wi - wy * nWx;
    const int py = (fr >> 1) & 1, px = fr & 1;
    half8_t xn[4][KS2];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      const int64_t row = ((int64_t)im * Gh + wy * 4 + 2 * py + (tb >> 1)) * Gw + wx * 4 + 2 * px + (tb & 1);
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) xn[tb][ks] = *reinterpret_cast<const half8_t*>(h + row * DI2 + ks * 32 + fg * 8);
    }
    auto step = [&]() -> const char* {
      const int left = ctotal - 1 - c;
      wait_tiles<PT2>(left < LA2 - 1 ? left : LA2 - 1);
      __builtin_amdgcn_s_barrier();
      if (c + LA2 < ctotal) issue(c + LA2);
      const char* m = smem + (c % NST2) * MAT;
      ++c;
      return m;
    };
    auto frag = [&](const char* m, const int row0, const int ks) {  // 512-byte rows
      return *reinterpret_cast<const half8_t*>(m + (row0 + fr) * 512 + ((((ks << 2) + fg) ^ fr) << 4));
    };

    // ---- the shortcut: proj(h) pooled (+ its bias + the output projection's): the accumulators of the block's output
    f32x4 accp[28];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const char* m = step();
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int rb = j * 4 + r4;
        f32x4 acc[4];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          const half8_t a = frag(m, r4 * 16, ks);
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[tb][ks], acc[tb], 0, 0, 0);
        }
        accp[rb] = vmax4(acc[0], acc[1], acc[2], acc[3]) + *reinterpret_cast<const f32x4*>(bias + rb * 16 + fg * 4);
      }
    }
#pragma unroll 1
    for (int hh = 0; hh < HEADS2; ++hh) {
      const float* bh = bias + DO2 + hh * 192;
      half8_t pf[2];
      {
        half8_t qf[2], kf[4][2];
#pragma unroll
        for (int sec = 0; sec < 2; ++sec) {  // image q, then image k of the head
          const char* m = step();
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            f32x4 acc[2][4];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
              for (int tb = 0; tb < 4; ++tb) acc[r2][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
              for (int ks = 0; ks < KS2; ++ks) {
                const half8_t a = frag(m, (2 * s + r2) * 16, ks);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) acc[r2][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[tb][ks], acc[r2][tb], 0, 0, 0);
              }
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bh + sec * 64 + (2 * s) * 16 + fg * 4);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bh + sec * 64 + (2 * s + 1) * 16 + fg * 4);
            if (sec == 0) {
              qf[s] = pack8(vmax4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]) + b0, vmax4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]) + b1);
            } else {
#pragma unroll
              for (int tb = 0; tb < 4; ++tb) kf[tb][s] = pack8(acc[0][tb] + b0, acc[1][tb] + b1);
            }
          }
        }
        // S^T[key block][16 queries of four windows]; a lane keeps its own window's keys
        f32x4 sacc[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          sacc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 2; ++s) sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][s], qf[s], sacc[kb], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[kb][r] = mine ? sacc[kb][r] : -INFINITY;
        }
        float ma = hmax3(sacc[0][0], sacc[0][1], sacc[0][2]);
        float mc = hmax3(sacc[2][0], sacc[2][1], sacc[2][2]);
        ma = hmax3(ma, sacc[0][3], sacc[1][0]);
        mc = hmax3(mc, sacc[2][3], sacc[3][0]);
        ma = hmax3(ma, sacc[1][1], sacc[1][2]);
        mc = hmax3(mc, sacc[3][1], sacc[3][2]);
        ma = hmax3(ma, sacc[1][3], sacc[3][3]);
        const float nmb = -(hrow_max4(ma, mc) * sl2);
        f32x4 e[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) e[kb][r] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], sl2, nmb));
#pragma unroll
        for (int s = 0; s < 2; ++s) pf[s] = pack8(e[2 * s], e[2 * s + 1]);
      }
      half8_t of[2];
      {  // image v: one d block at a time, each straight into O^T
        const char* m = step();
        f32x4 oacc[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const float b = bh[128 + db * 16 + fr];
          f32x4 acc[4];
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[tb] = f32x4{b, b, b, b};
#pragma unroll
          for (int ks = 0; ks < KS2; ++ks) {
            const half8_t bw = frag(m, db * 16, ks);
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xn[tb][ks], bw, acc[tb], 0, 0, 0);
          }
          oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(acc[0], acc[1]), pf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(acc[2], acc[3]), pf[1], oacc[db], 0, 0, 0);
        }
        const float l = __shfl(oacc[3][3], 48 + fr, 64);
        const float inv = 1.0f / l;
#pragma unroll
        for (int s = 0; s < 2; ++s) of[s] = pack8(oacc[2 * s] * inv, oacc[2 * s + 1] * inv);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // the projection's columns of the head, output rows 0..223 then 224..447 (128-byte rows)
        const char* m = step();
#pragma unroll
        for (int o14 = 0; o14 < 14; ++o14)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const half8_t a = *reinterpret_cast<const half8_t*>(m + (o14 * 16 + fr) * 128 + ((((s << 2) + fg) ^ ((fr >> 1) & 7)) << 4));
            accp[half * 14 + o14] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, of[s], accp[half * 14 + o14], 0, 0, 0);
          }
      }
    }
    if (live) {
      const int64_t orow = ((int64_t)im * (Gh >> 1) + wy * 2 + py) * Go + wx * 2 + px;
#pragma unroll
      for (int ob = 0; ob < 28; ++ob) *reinterpret_cast<f32x4*>(out + orow * DO2 + ob * 16 + fg * 4) = accp[ob];
    }
  }
}

}  // namespace

// h f16 [n_img * Gh * Gw, Din] contiguous = layer_norm1(x) on the (Gh x Gw) grid; out f32 [n_img * Gh/2 * Gw/2, Dout] contiguous: the
// block's hidden state after the attention half, on the pooled grid.  112 -> 224 (4 heads, 8 x 8 windows: Gh, Gw multiples of 8):
// w_img f16 [14][16384], bias f32 [224 + 4 * 192 + 224]; 224 -> 448 (8 heads, 4 x 4 windows): w_img [47][16384], bias [448 + 8 * 192]
// (lmx/sam.py pack_hiera_attn_pool).
extern "C" int lmx_k_hiera_attn_pool(const void* h, float* out, const void* w_img, const float* bias, int n_img, int Gh, int Gw, int Din,
                                     int Dout, int heads, float scale, lmx_stream_t stream) {
  LMX_REQUIRE(h && out && w_img && bias, "lmx_k_hiera_attn_pool: null pointer");
  const bool s3 = Din == DI2 && Dout == DO2 && heads == HEADS2;  // the block that opens stage 3: 4 x 4 windows
  LMX_REQUIRE(s3 || (Din == DI && Dout == DO && heads == HEADSP),
              "lmx_k_hiera_attn_pool: built for 112 -> 224 channels with 4 heads and 224 -> 448 with 8, got %d -> %d, %d heads", Din, Dout, heads);
  const int ws = s3 ? 4 : 8;
  LMX_REQUIRE(n_img > 0 && Gh > 0 && Gw > 0 && Gh % ws == 0 && Gw % ws == 0, "lmx_k_hiera_attn_pool: token grid %d x %d is not whole %d x %d windows", Gh, Gw, ws, ws);
  LMX_REQUIRE(aligned16(h) && aligned16(out) && aligned16(w_img), "lmx_k_hiera_attn_pool: alignment");
  const int64_t nwin = (int64_t)n_img * (Gh / ws) * (Gw / ws);
  LMX_REQUIRE(nwin < (1ll << 31), "lmx_k_hiera_attn_pool: too many windows");
  const int per = s3 ? NW2 * 4 : NWP;  // windows per workgroup pass
  const int64_t ngroup = (nwin + per - 1) / per;
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera_attnp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEMP));
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera_attnq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2));
    attr_set = true;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    LMX_HIP(hipGetDevice(&dev));
    LMX_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const unsigned grid = (unsigned)(ngroup < n_cu ? ngroup : n_cu);
  if (s3)
    hipLaunchKernelGGL(hiera_attnq_kernel, dim3(grid), dim3(NW2 * 64), SMEM2, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const half_t*>(h), out, reinterpret_cast<const half_t*>(w_img), bias, Gh, Gw, (int)nwin, (int)ngroup,
                       scale * 1.44269504088896340736f);
  else
    hipLaunchKernelGGL(hiera_attnp_kernel, dim3(grid), dim3(NWP * 64), SMEMP, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const half_t*>(h), out, reinterpret_cast<const half_t*>(w_img), bias, Gh, Gw, (int)nwin, (int)ngroup,
                       scale * 1.44269504088896340736f);
  return lmx_launch_check("hiera_attn_pool kernel");
}

// ================================================================================================================================
// The MLP half of the same blocks in the same form (round 3): x += fc2(gelu(fc1(layer_norm2(x)))) for D = 112 / 224 with the weights
// streamed as host-made LDS images in steps of 64 hidden units.  What csrc/mlp.hip's kernel (which this replaces for these widths)
// pays for: it streams all 8 D^2 bytes of weights through LDS per 128 tokens in 32-unit chunks — 3.4 GB of L2 -> LDS traffic per
// 30-frame launch at either width, a barrier per 28 - 30 MFMAs of a wave (profiles/r03_fused_mlp_decompose.txt: two thirds of its
// time is that fixed cost).  Here a workgroup is 8 waves x 32 (D = 224) or 48 (D = 112) tokens per pass over the weights and a step
// is 56 - 90 MFMAs per wave.  It pays from a few hundred thousand rows: 1123 -> 966 us at 1.97 M rows of D = 112, 825 -> 736 us at
// 491 520 rows of D = 224 (tools/mlp_probe.py, with the next block's LayerNorm output), but 346 -> 401 and 236 -> 314 us at a third
// of those rows — 256 persistent workgroups with 3 - 7 passes each start and drain badly.  lmx/sam.py uses it at every batch size all
// the same: the two kernels sum in different orders, and a frame's result must not depend on the batch it rides in.  (Tried on top, no change: fc2 of step c - 1 issued under the GELU of step c
// inside the wave — 959 / 731 us against 952 - 966 / 736 - 750: the kernel is not waiting for its matrix and vector work to overlap.)  A wave reads its 32 f32 rows once in accumulator layout (lane (token fr, fg) holds features 16 ob + 4 fg .. + 3):
// they are layer_norm2's input (statistics over the four lanes of a token), and + b2 the initial fc2 accumulators; the normalised
// rows, rounded to f16, are fc1's B operand in k-slot order (W1's columns are stored in that order), fc1's GELU'd accumulators are
// fc2's B operand (W2's columns of a step likewise).  Outputs as lmx_k_ln_mlp: x in place, optionally its f16 copy and the next
// block's layer_norm1 rows.  Rounding points as in csrc/mlp.hip (LayerNorm output and GELU output f16, sums f32, rsq for the
// LayerNorm's 1 / sqrt).
namespace {

template <int DD>
struct MlpCfg {
  static constexpr int KS = (DD + 31) / 32;            // k-steps over the token width (112 -> 4, the last one half empty)
  static constexpr int OB = DD / 16;                   // 16-feature blocks of a row
  static constexpr int NCH = 4 * DD / 64;              // steps of 64 hidden units
  static constexpr bool ONE = DD <= 128;               // W1 step (64 rows x 256 B) and W2 step (DD rows x 128 B) share one 32 KB image
  static constexpr int NIMG = ONE ? NCH : 2 * NCH;
  static constexpr int ROW1 = DD <= 128 ? 256 : 512;   // bytes per W1 image row
  static constexpr int NB = 4 * DD + 5 * DD;           // b1 [4D], b2 [D], gamma2, beta2, gamma_next, beta_next [D each]
};
#ifndef LMX_MLP_TB112
#define LMX_MLP_TB112 3  // measured at 1.97 M rows: 2 blocks 1006 us, 3 blocks 966 us, 4 blocks 998 us (csrc/mlp.hip: 1123 - 1173 us)
#endif
constexpr int NWM = 8, NSTM = 4, LAM = NSTM - 1, PTM = MAT / (NWM * 1024);
template <int DD>
constexpr int mlp_tb() { return DD == 112 ? LMX_MLP_TB112 : 2; }  // 16-token blocks per wave

template <int DD>
__global__ __launch_bounds__(NWM * 64, 2) void hiera_mlp_kernel(float* __restrict__ x, const int64_t ldx, const half_t* __restrict__ img,
                                                                 const float* __restrict__ bias_g, const float eps, const int64_t rows,
                                                                 half_t* __restrict__ x16, half_t* __restrict__ h_n, const int ngroup) {
  using C = MlpCfg<DD>;
  constexpr int KS = C::KS, OB = C::OB, TBM = mlp_tb<DD>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias = reinterpret_cast<float*>(smem + NSTM * MAT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  for (int i = tid; i < C::NB; i += NWM * 64) bias[i] = bias_g[i];
  const float* b1s = bias;
  const float* b2s = bias + 4 * DD;
  const float* g2s = bias + 5 * DD;
  const float* e2s = bias + 6 * DD;
  const float* gns = bias + 7 * DD;
  const float* ens = bias + 8 * DD;

  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(img), 0, C::NIMG * MAT, 0x00020000);
  const int my_groups = (ngroup - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ctotal = my_groups * C::NIMG;
  const unsigned voff = (unsigned)(wave * (PTM * 1024) + lane * 16);
  auto issue = [&](const int c) {
    char* dst = smem + (c % NSTM) * MAT + wave * (PTM * 1024);
#pragma unroll
    for (int t = 0; t < PTM; ++t) lds_dma16(w_rs, dst + t * 1024, voff + t * 1024, (c % C::NIMG) * MAT);
  };
#pragma unroll
  for (int c = 0; c < LAM; ++c)
    if (c < ctotal) issue(c);
  __syncthreads();  // the biases and LayerNorm vectors are in LDS for every wave

  int c = 0;
  for (int grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
    int64_t trow[TBM];
    bool live[TBM];
#pragma unroll
    for (int tb = 0; tb < TBM; ++tb) {
      const int64_t t = (int64_t)grp * (NWM * TBM * 16) + wave * (TBM * 16) + tb * 16 + fr;
      live[tb] = t < rows;
      trow[tb] = live[tb] ? t : 0;
    }
    // the rows in accumulator layout; layer_norm2 over the four lanes of a token
    f32x4 oacc[OB][TBM];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int tb = 0; tb < TBM; ++tb) oacc[ob][tb] = *reinterpret_cast<const f32x4*>(x + trow[tb] * ldx + ob * 16 + fg * 4);
    half8_t xn[TBM][KS];
    {
      float mean[TBM], rstd[TBM];
#pragma unroll
      for (int tb = 0; tb < TBM; ++tb) {
        float sm = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) sm += (oacc[ob][tb][0] + oacc[ob][tb][1]) + (oacc[ob][tb][2] + oacc[ob][tb][3]);
        mean[tb] = hrow_sum4(sm) * (1.0f / DD);
        float sq = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
          const f32x4 dl = oacc[ob][tb] - mean[tb];
          sq += (dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]);
        }
        rstd[tb] = __builtin_amdgcn_rsqf(hrow_sum4(sq) * (1.0f / DD) + eps);
      }
      // two feature blocks (one k-step of fc1) at a time: the normalised values are short-lived
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f32x4 nrm[2][TBM];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int ob = 2 * ks + hb;
          if (ob < OB) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(g2s + ob * 16 + fg * 4), b = *reinterpret_cast<const f32x4*>(e2s + ob * 16 + fg * 4);
#pragma unroll
            for (int tb = 0; tb < TBM; ++tb) nrm[hb][tb] = (oacc[ob < OB ? ob : 0][tb] - mean[tb]) * rstd[tb] * g + b;
          } else {
#pragma unroll
            for (int tb = 0; tb < TBM; ++tb) nrm[hb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int tb = 0; tb < TBM; ++tb) xn[tb][ks] = pack8(nrm[0][tb], nrm[1][tb]);
      }
    }
    // the fc2 accumulators start from x + b2
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b2s + ob * 16 + fg * 4);
#pragma unroll
      for (int tb = 0; tb < TBM; ++tb) oacc[ob][tb] += bv;
    }
    auto step = [&]() -> const char* {
      const int left = ctotal - 1 - c;
      wait_tiles<PTM>(left < LAM - 1 ? left : LAM - 1);
      __builtin_amdgcn_s_barrier();
      if (c + LAM < ctotal) issue(c + LAM);
      const char* m = smem + (c % NSTM) * MAT;
      ++c;
      return m;
    };
#pragma unroll 1
    for (int ch = 0; ch < C::NCH; ++ch) {
      const char* m1 = step();
      // ---- H^T[64 hidden][tokens] = W1 step . LN(x)^T (+ b1), GELU, rounded: fc2's B operand in k-slot order
      half8_t pf[TBM][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 acc[2][TBM];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + ch * 64 + (2 * s + r2) * 16 + fg * 4);
#pragma unroll
          for (int tb = 0; tb < TBM; ++tb) acc[r2][tb] = bv;
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const half8_t a = *reinterpret_cast<const half8_t*>(m1 + ((2 * s + r2) * 16 + fr) * C::ROW1 + ((((ks << 2) + fg) ^ fr) << 4));
#pragma unroll
            for (int tb = 0; tb < TBM; ++tb) acc[r2][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[tb][ks], acc[r2][tb], 0, 0, 0);
          }
#pragma unroll
        for (int tb = 0; tb < TBM; ++tb) {
          const f32x2 g0 = gelu_pk(f32x2{acc[0][tb][0], acc[0][tb][1]}), g1 = gelu_pk(f32x2{acc[0][tb][2], acc[0][tb][3]});
          const f32x2 g2 = gelu_pk(f32x2{acc[1][tb][0], acc[1][tb][1]}), g3 = gelu_pk(f32x2{acc[1][tb][2], acc[1][tb][3]});
          pf[tb][s] = half8_t{(half_t)g0[0], (half_t)g0[1], (half_t)g1[0], (half_t)g1[1], (half_t)g2[0], (half_t)g2[1], (half_t)g3[0], (half_t)g3[1]};
        }
      }
      // ---- x^T[o][token] += W2[:, step] . H   (image rows of 128 B: chunk c of row r at c ^ ((r >> 1) & 7))
      const char* m2 = C::ONE ? m1 + 16384 : step();
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const half8_t a = *reinterpret_cast<const half8_t*>(m2 + (ob * 16 + fr) * 128 + ((((s << 2) + fg) ^ ((fr >> 1) & 7)) << 4));
#pragma unroll
          for (int tb = 0; tb < TBM; ++tb) oacc[ob][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[tb][s], oacc[ob][tb], 0, 0, 0);
        }
    }
    // ---- outputs: x; its f16 copy; the next block's layer_norm1 rows
#pragma unroll
    for (int tb = 0; tb < TBM; ++tb) {
      if (live[tb]) {
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) *reinterpret_cast<f32x4*>(x + trow[tb] * ldx + ob * 16 + fg * 4) = oacc[ob][tb];
        if (x16) {
#pragma unroll
          for (int ob = 0; ob < OB; ++ob) {
            const f32x4 v = oacc[ob][tb];
            *reinterpret_cast<half4_t*>(x16 + trow[tb] * DD + ob * 16 + fg * 4) = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          }
        }
      }
      if (h_n) {
        float sm = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) sm += (oacc[ob][tb][0] + oacc[ob][tb][1]) + (oacc[ob][tb][2] + oacc[ob][tb][3]);
        const float mean = hrow_sum4(sm) * (1.0f / DD);
        float sq = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
          const f32x4 a = oacc[ob][tb] - mean;
          sq += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
        }
        const float rstd = __builtin_amdgcn_rsqf(hrow_sum4(sq) * (1.0f / DD) + eps);
        if (live[tb]) {
#pragma unroll
          for (int ob = 0; ob < OB; ++ob) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gns + ob * 16 + fg * 4), b = *reinterpret_cast<const f32x4*>(ens + ob * 16 + fg * 4);
            const f32x4 v = (oacc[ob][tb] - mean) * rstd * g + b;
            *reinterpret_cast<half4_t*>(h_n + trow[tb] * DD + ob * 16 + fg * 4) = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          }
        }
      }
    }
  }
}

template <int DD>
int launch_hiera_mlp(float* x, int64_t ldx, const void* img, const float* bias, float eps, int64_t rows, void* x16, void* h_next, hipStream_t st) {
  constexpr int SM = NSTM * MAT + MlpCfg<DD>::NB * 4, TBM = mlp_tb<DD>();
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera_mlp_kernel<DD>), hipFuncAttributeMaxDynamicSharedMemorySize, SM));
    attr_set = true;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    LMX_HIP(hipGetDevice(&dev));
    LMX_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int64_t ngroup = (rows + NWM * TBM * 16 - 1) / (NWM * TBM * 16);
  LMX_REQUIRE(ngroup < (1ll << 31), "lmx_k_ln_mlp_img: too many rows");
  const unsigned grid = (unsigned)(ngroup < n_cu ? ngroup : n_cu);
  hipLaunchKernelGGL(hiera_mlp_kernel<DD>, dim3(grid), dim3(NWM * 64), SM, st, x, ldx, reinterpret_cast<const half_t*>(img), bias, eps, rows,
                     reinterpret_cast<half_t*>(x16), reinterpret_cast<half_t*>(h_next), (int)ngroup);
  return lmx_launch_check("hiera_mlp_kernel");
}

}  // namespace

// x f32 [rows, ldx] updated in place: x += fc2(gelu(fc1(LayerNorm(x)))), D = 112 or 224.  w_img f16 [NIMG][16384], bias f32
// [b1 (4D) | b2 | gamma2 | beta2 | gamma_next | beta_next (D each)]: lmx/sam.py pack_ln_mlp.  x16 / h_next as in lmx_k_ln_mlp.
extern "C" int lmx_k_ln_mlp_img(float* x, int64_t ldx, const void* w_img, const float* bias, int64_t rows, int D_, float eps, void* x16,
                                void* h_next, lmx_stream_t stream) {
  LMX_REQUIRE(x && w_img && bias, "lmx_k_ln_mlp_img: null pointer");
  LMX_REQUIRE(D_ == 112 || D_ == 224, "lmx_k_ln_mlp_img: D=%d (built for 112 and 224)", D_);
  LMX_REQUIRE(rows > 0 && ldx >= D_ && ldx % 4 == 0 && aligned16(x) && aligned16(w_img), "lmx_k_ln_mlp_img: rows / ldx / alignment");
  LMX_REQUIRE((((uintptr_t)x16) & 7) == 0 && (((uintptr_t)h_next) & 7) == 0, "lmx_k_ln_mlp_img: x16 / h_next must be 8-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (D_ == 112) return launch_hiera_mlp<112>(x, ldx, w_img, bias, eps, rows, x16, h_next, st);
  return launch_hiera_mlp<224>(x, ldx, w_img, bias, eps, rows, x16, h_next, st);
}
