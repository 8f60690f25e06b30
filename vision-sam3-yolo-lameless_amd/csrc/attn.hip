// attn.hip — flash-style attention for the ViT blocks (DINO 201/257 tokens, SAM/Hiera windows and 4096-token
// global blocks).  Replaces eager_attention_forward (TF:models/sam2/modeling_sam2.py:262-288,
// TF:models/dinov3_vit/modeling_dinov3_vit.py:203-235) and the window_partition / window_unpartition copies
// around it (TF:models/sam2/modeling_sam2.py:412-455): windows are addressed in place on the [Gh][Gw] token grid.
//
// Per workgroup: one (batch element | window, head) and 64*QB queries; 4 waves x (16*QB) queries each.
// Per 64-key tile (K and V^T staged in LDS, 8 KB each):
//   S^T = K . Q^T   on v_mfma_f32_16x16x32_f16 with a = K fragment, b = Q fragment  ->  D[key][query]:
//         a lane holds, for ITS query (lane&15), the 16 keys {16*kb + 4*(lane>>4) + r};  the row softmax is a
//         16-value local reduction + two __shfl_xor steps (16, 32).
//   O^T += V^T . P^T with a = V^T fragment, b = P^T fragment.  The MFMA k index is a free permutation as long as
//         both operands agree, so k-slot (ks, g, j) is bound to key 32*ks + 16*(j>>2) + 4*g + (j&3): the P^T
//         fragment is then exactly the lane's own S^T accumulators (no cross-lane movement, no LDS round trip;
//         cdna guide §3 "An accumulator tile as the next MFMA's operand"), and V is written to LDS transposed in
//         that slot order so its fragment is one ds_read_b128.
// Both LDS images have 128-B rows with the 16-B chunk index XOR-swizzled by (row&7) (conflict-free b128 reads).
#include "common.h"

namespace {

struct Geo {
  int mode, Tq, Tk, Gh, Gw, ws, nWy, nWx, Gqh, Gqw, wsq;
};

__device__ __forceinline__ int64_t key_row(const Geo& g, int b, int t) {
  if (g.mode == 0) return (int64_t)b * g.Tk + t;
  const int nW = g.nWy * g.nWx;
  const int img = b / nW;
  const int w = b - img * nW;
  const int wy = w / g.nWx, wx = w - wy * g.nWx;
  const int ty = t / g.ws, tx = t - ty * g.ws;
  const int y = wy * g.ws + ty, x = wx * g.ws + tx;
  if (y >= g.Gh || x >= g.Gw) return -1;
  return ((int64_t)img * g.Gh + y) * g.Gw + x;
}
__device__ __forceinline__ int64_t query_row(const Geo& g, int b, int t) {
  if (g.mode == 0) return (int64_t)b * g.Tq + t;
  const int nW = g.nWy * g.nWx;
  const int img = b / nW;
  const int w = b - img * nW;
  const int wy = w / g.nWx, wx = w - wy * g.nWx;
  const int ty = t / g.wsq, tx = t - ty * g.wsq;
  const int y = wy * g.wsq + ty, x = wx * g.wsq + tx;
  if (y >= g.Gqh || x >= g.Gqw) return -1;
  return ((int64_t)img * g.Gqh + y) * g.Gqw + x;
}

template <int QB>
__global__ __launch_bounds__(256) void attn_kernel(const lmx_attn_desc p, const Geo geo, const int nQT) {
  __shared__ __attribute__((aligned(16))) half_t Ks[64 * 64];
  __shared__ __attribute__((aligned(16))) half_t Vt[64 * 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;

  int bid = blockIdx.x;
  const int qt = bid % nQT;
  bid /= nQT;
  const int h = bid % p.H;
  const int b = bid / p.H;

  const half_t* Q = reinterpret_cast<const half_t*>(p.Q);
  const half_t* K = reinterpret_cast<const half_t*>(p.K);
  const half_t* V = reinterpret_cast<const half_t*>(p.V);
  const half_t* padk = reinterpret_cast<const half_t*>(p.pad_k);
  const half_t* padv = reinterpret_cast<const half_t*>(p.pad_v);
  half_t* O = reinterpret_cast<half_t*>(p.O);
  const int hd = p.hd;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- Q fragments (B operand): lane holds Q[q = fr][d = 32*ks + 8*fg + j]
  const int q_base = qt * (64 * QB) + wave * (16 * QB);
  half8_t qf[QB][2];
  int64_t qrow[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = q_base + qb * 16 + fr;
    qrow[qb] = (tq < p.Tq) ? query_row(geo, b, tq) : -1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int d = ks * 32 + fg * 8;
      qf[qb][ks] = (qrow[qb] >= 0 && d < hd)
                       ? *reinterpret_cast<const half8_t*>(Q + qrow[qb] * p.ldq + (int64_t)h * hd + d)
                       : zero8;
    }
  }

  f32x4 oacc[QB][4];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging assignment: chunk c of key kk = (tid>>3) + 32*i
  const int sc = tid & 7;
  const int sk = tid >> 3;
  half8_t kst[2], vst[2];
  auto load_tile = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = t0 + sk + 32 * i;
      const int d = sc * 8;
      kst[i] = zero8;
      vst[i] = zero8;
      if (t < p.Tk && d < hd) {
        const int64_t row = key_row(geo, b, t);
        if (row >= 0) {
          kst[i] = *reinterpret_cast<const half8_t*>(K + row * p.ldk + (int64_t)h * hd + d);
          vst[i] = *reinterpret_cast<const half8_t*>(V + row * p.ldv + (int64_t)h * hd + d);
        } else {
          if (padk) kst[i] = *reinterpret_cast<const half8_t*>(padk + (int64_t)h * hd + d);
          if (padv) vst[i] = *reinterpret_cast<const half8_t*>(padv + (int64_t)h * hd + d);
        }
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = sk + 32 * i;
      *reinterpret_cast<half8_t*>(Ks + kk * 64 + ((sc ^ (kk & 7)) << 3)) = kst[i];
      // V^T in k-slot order: key kk -> slot = 32*ks + 8*g + 4*jhi + jlo
      const int ks = kk >> 5, rem = kk & 31, jhi = rem >> 4, g = (rem & 15) >> 2, jlo = rem & 3;
      const int chunk = ks * 4 + g, within = jhi * 4 + jlo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = sc * 8 + e;
        Vt[d * 64 + ((chunk ^ (d & 7)) << 3) + within] = vst[i][e];
      }
    }
  };

  const float sl2 = p.scale * 1.44269504088896340736f;  // scores are kept in log2 units
  const int ntile = (p.Tk + 63) / 64;
  load_tile(0);
  for (int it = 0; it < ntile; ++it) {
    const int t0 = it * 64;
    __syncthreads();  // previous tile's readers are done
    store_tile();
    __syncthreads();
    if (it + 1 < ntile) load_tile(t0 + 64);

    // ---- S^T = K . Q^T
    f32x4 sacc[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) sacc[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = (((ks << 2) + fg) ^ (fr & 7)) << 3;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (kb * 16 + fr) * 64 + coff);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          sacc[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qb][ks], sacc[qb][kb], 0, 0, 0);
      }
    }

    // ---- online softmax (per query = per lane column), P^T fragments straight from the accumulators
    half8_t pf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t0 + kb * 16 + fg * 4 + r;
          float s = sacc[qb][kb][r] * sl2;
          s = key < p.Tk ? s : -INFINITY;
          sacc[qb][kb][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
      m_run[qb] = m_new;
      float rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(sacc[qb][kb][r] - m_new);
          const half_t eh = (half_t)e;
          rs += (float)eh;  // the normaliser sums exactly what the PV MFMA sees
          pf[qb][kb >> 1][(kb & 1) * 4 + r] = eh;
        }
      l_run[qb] = l_run[qb] * alpha + rs;
#pragma unroll
      for (int db = 0; db < 4; ++db) oacc[qb][db] *= alpha;
    }

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int d = db * 16 + fr;
        const half8_t vf = *reinterpret_cast<const half8_t*>(Vt + d * 64 + ((((ks << 2) + fg) ^ (d & 7)) << 3));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qb][ks], oacc[qb][db], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane owns O[q = fr][d = 16*db + 4*fg + r]
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qrow[qb] < 0) continue;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int d = db * 16 + fg * 4;
      if (d >= hd) continue;
      half4_t o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (half_t)(oacc[qb][db][r] * inv);
      *reinterpret_cast<half4_t*>(O + qrow[qb] * p.ldo + (int64_t)h * hd + d) = o;
    }
  }
}

}  // namespace

extern "C" int lmx_k_attention(const lmx_attn_desc* dp, lmx_stream_t stream) {
  LMX_REQUIRE(dp != nullptr, "lmx_k_attention: null descriptor");
  const lmx_attn_desc& d = *dp;
  LMX_REQUIRE(d.Q && d.K && d.V && d.O, "lmx_k_attention: null Q/K/V/O");
  LMX_REQUIRE(d.B > 0 && d.H > 0 && d.Tq > 0 && d.Tk > 0, "lmx_k_attention: empty problem");
  LMX_REQUIRE(d.hd % 8 == 0 && d.hd > 0 && d.hd <= 64, "lmx_k_attention: head dim %d (need multiple of 8, <=64)", d.hd);
  LMX_REQUIRE(d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 4 == 0, "lmx_k_attention: strides");
  LMX_REQUIRE(aligned16(d.Q) && aligned16(d.K) && aligned16(d.V) && ((((uintptr_t)d.O) & 7) == 0),
              "lmx_k_attention: alignment");
  Geo g{};
  g.mode = d.mode;
  g.Tq = d.Tq;
  g.Tk = d.Tk;
  if (d.mode == 1) {
    LMX_REQUIRE(d.ws > 0 && d.Gh > 0 && d.Gw > 0, "lmx_k_attention: window geometry");
    const int qs = d.q_stride > 0 ? d.q_stride : 1;
    LMX_REQUIRE(qs == 1 || qs == 2, "lmx_k_attention: q_stride %d", qs);
    LMX_REQUIRE(d.ws % qs == 0 && d.Gh % qs == 0 && d.Gw % qs == 0, "lmx_k_attention: q_stride must divide ws/Gh/Gw");
    g.Gh = d.Gh;
    g.Gw = d.Gw;
    g.ws = d.ws;
    g.nWy = (d.Gh + d.ws - 1) / d.ws;
    g.nWx = (d.Gw + d.ws - 1) / d.ws;
    g.Gqh = d.Gh / qs;
    g.Gqw = d.Gw / qs;
    g.wsq = d.ws / qs;
    LMX_REQUIRE(d.Tk == d.ws * d.ws && d.Tq == g.wsq * g.wsq, "lmx_k_attention: Tq/Tk do not match window size");
    LMX_REQUIRE(d.B % (g.nWy * g.nWx) == 0, "lmx_k_attention: B=%d not a multiple of windows per image %d", d.B,
                g.nWy * g.nWx);
    if (d.pad_k) LMX_REQUIRE(aligned16(d.pad_k), "lmx_k_attention: pad_k alignment");
    if (d.pad_v) LMX_REQUIRE(aligned16(d.pad_v), "lmx_k_attention: pad_v alignment");
  } else {
    LMX_REQUIRE(d.mode == 0, "lmx_k_attention: bad mode %d", d.mode);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool big = d.Tq > 64;
  const int qtile = big ? 128 : 64;
  const int nQT = (d.Tq + qtile - 1) / qtile;
  const int64_t nblk = (int64_t)d.B * d.H * nQT;
  LMX_REQUIRE(nblk < (1ll << 31), "lmx_k_attention: grid too large");
  if (big)
    hipLaunchKernelGGL((attn_kernel<2>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  else
    hipLaunchKernelGGL((attn_kernel<1>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  return lmx_launch_check("attn_kernel");
}
