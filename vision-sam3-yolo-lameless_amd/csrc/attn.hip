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
//         cdna guide §3 "An accumulator tile as the next MFMA's operand").  V stays ROW-major in LDS (coalesced
//         16-byte staging like K) and its fragment is two ds_read_b64_tr_b16 (hardware transpose, guide T10): the
//         4 consecutive keys of each half-fragment are exactly one 4 x 16 transposed block.
// Softmax is the VALU bottleneck at head dim 64 (16 MFMA cycles per score element-lane vs ~5 VALU + 1 exp), so:
// scores stay raw and the scale is folded into one fma per element (exp2(s*c - m*c)); masking runs only on the last
// tile; probabilities are packed with v_cvt_pk_f16_f32; the row sum is either free (head dim <= 56: V's unused
// column 63 is set to 1, the PV MFMA accumulates the normaliser) or one v_dot2 per pair.
// Both LDS images have 128-B rows with the 16-B chunk index XOR-swizzled by (row&7) (conflict-free reads), and are
// double-buffered: one barrier per 64-key tile.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#ifdef LMX_DBG_TIMELINE
// development build only (make O=obj_tl EXTRA=-DLMX_DBG_TIMELINE LIB=../lmx/liblmx_tl.so; tools/attn_spp_timeline.py): per item of
// workgroup 0, every wave stamps s_memtime at the phase boundaries of attn_spp_kernel
__device__ unsigned long long lmx_attn_tl[64 * 8 * 8];  // [item][wave][stamp]
extern "C" int lmx_dbg_get_attn_timeline(void* host, int64_t bytes) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lmx_attn_tl), bytes); }
#define ATL(i) if (blockIdx.x == 0 && lane == 0 && tl_it < 64) lmx_attn_tl[(tl_it * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter()
#else
#define ATL(i)
#endif
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

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4_t* lds_f16x4_ptr;

// ds_read_b64_tr_b16: per 16-lane group, a 4(rows) x 16(cols) block of halfs is delivered column-major (lane i gets
// column i of the 4 rows).  Needs EXEC all ones and 8-byte aligned addresses (cdna guide §5.5 T10).
__device__ __forceinline__ half4_t lds_tr_read(const half_t* p) {
  const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4_ptr)(p));
  half4_t h;
  __builtin_memcpy(&h, &v, 8);
  return h;
}

// LDS reads at `base + compile-time offset` with the base made opaque to the optimiser: hipcc's loop strength reduction
// otherwise rewrites the ring-slot addressing into bases with NEGATIVE constant parts, which cannot be DS offset immediates,
// and spends ~45 integer VALU per key tile on addresses (a quarter of the VALU work of a tile that is VALU-bound).
typedef __attribute__((address_space(3))) const half8_t* lds_h8_cptr;
__device__ __forceinline__ unsigned lds_addr(const half_t* p) {
  unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const half_t*)p;
  asm volatile("" : "+v"(a));
  return a;
}
__device__ __forceinline__ half8_t lds_read8(unsigned base, int byte_off) { return *(lds_h8_cptr)(uintptr_t)(base + (unsigned)byte_off); }
__device__ __forceinline__ half4_t lds_tr_read_at(unsigned base, int byte_off) {
  const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4_ptr)(uintptr_t)(base + (unsigned)byte_off));
  half4_t h;
  __builtin_memcpy(&h, &v, 8);
  return h;
}

// QB: 16-query blocks per wave.  ONES: head dim <= 56, so V's (zero) column 63 is set to 1 and the PV MFMA itself
// accumulates the softmax normaliser in O[:,63] — no row-sum instructions at all.
// REL: SAM v1 decomposed relative-position bias read from the per-query tables of relpos_tables_kernel.
// DMA (flat geometry only): K/V tiles arrive by LDS-DMA (buffer_load ... lds) into a 3-slot ring, two tiles in flight, a
// counted s_waitcnt vmcnt + one raw s_barrier per tile.  The register-staged path loads a tile at the top of an
// iteration and stores it to LDS at the bottom of the SAME iteration: with 64 key tiles per query block (Hiera global
// attention, T = 4096) every iteration then waits out an L2/HBM latency.
// HDW: head-dim class.  64: LDS rows of 64 halfs, 2 k-steps for S, 4 output blocks.  96 (SAM ViT-H, head dim 80): LDS rows of
// 128 halfs (16 chunks, swizzle over row & 15), 3 k-steps, 6 output blocks — the layout and the MFMA bookkeeping are the
// same, only wider.
template <int QB, bool ONES, bool REL, bool DMA, int HDW>
__global__ __launch_bounds__(256, 2) void attn_kernel(const lmx_attn_desc p, const Geo geo, const int nQT) {
#ifndef LMX_ATTN_RING
#define LMX_ATTN_RING 3  // LDS-DMA ring slots of 16 KB (K + V tile).  Measured: 3 slots 586 TFLOP/s, 4 and 5 slots 480-530 (profiles/r02_ab_attn_ring.txt)
#endif
  constexpr int NSL = DMA ? LMX_ATTN_RING : 2;
  constexpr int KS = HDW / 32;            // 32-wide k-steps of S = K . Q^T
  constexpr int NDB = HDW / 16;           // 16-wide blocks of the output head dim
  constexpr int RW = HDW <= 64 ? 64 : 128;  // halfs per LDS row
  constexpr int CPR = RW / 8;             // 16-byte chunks per row; XOR swizzle over row & (CPR - 1)
  constexpr int NPASS = 64 * CPR / 256;   // staging passes of 256 threads over a 64-key tile
  static_assert(!(DMA && HDW != 64) && !(ONES && HDW != 64), "LDS-DMA ring and the ones column are built for 64-wide rows");
  __shared__ __attribute__((aligned(16))) half_t Ks[NSL][64 * RW];
  __shared__ __attribute__((aligned(16))) half_t Vs[NSL][64 * RW];  // row-major [key][d], same swizzle as K
  static_assert(!(DMA && REL), "the relative-position bias is read from global memory inside the loop: vmcnt would not count DMAs only");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;

  // XCD-aware bijective remap: the hardware deals consecutive workgroups round-robin over the 8 XCDs; give each XCD a
  // contiguous range of work items instead, so the query tiles of one (image, head) share that XCD's L2 copy of K/V.
  int bid;
  {
    const int n = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = n >> 3, r = n & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nQTa = nQT < 0 ? -nQT : nQT;  // development switch: a negative count turns the lazy rescale off
  const int qt = bid % nQTa;
  bid /= nQTa;
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
  half8_t qf[QB][KS];
  int64_t qrow[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = q_base + qb * 16 + fr;
    qrow[qb] = (tq < p.Tq) ? query_row(geo, b, tq) : -1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d = ks * 32 + fg * 8;
      const bool ok = qrow[qb] >= 0 && d < hd;
      const half8_t qv = *reinterpret_cast<const half8_t*>(Q + (ok ? qrow[qb] * p.ldq + (int64_t)h * hd + d : 0));
      qf[qb][ks] = ok ? qv : zero8;
    }
  }

  // REL: this lane's query row of the bias table (2*rel_S halfs): [0,S) by key row, [S,2S) by key column
  const half_t* relq[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = q_base + qb * 16 + fr;
    relq[qb] = REL ? reinterpret_cast<const half_t*>(p.rel) + (((int64_t)b * p.H + h) * p.Tq + (tq < p.Tq ? tq : 0)) * (2 * p.rel_S)
                   : nullptr;
  }

  f32x4 oacc[QB][NDB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < NDB; ++db) oacc[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging assignment: chunk sc of key kk = sk + 32*i
  const int sc = tid & (CPR - 1);
  const int sk = tid / CPR;  // 32 (or 16) keys per pass
  constexpr int KPP = 256 / CPR;
  half8_t kst[NPASS], vst[NPASS];
  // per-thread base pointers: everything that does not depend on the key index is hoisted out of the tile loop
  const int64_t hoff0 = (int64_t)h * hd + sc * 8;
  const half_t* Kb = K + hoff0 + (geo.mode == 0 ? (int64_t)b * p.Tk * p.ldk : 0);
  const half_t* Vb = V + hoff0 + (geo.mode == 0 ? (int64_t)b * p.Tk * p.ldv : 0);
  const bool d_ok = sc * 8 < hd;
  auto load_tile = [&](int t0) {
    if (geo.mode == 0) {  // flat geometry (wave-uniform branch): row = b*Tk + t, no padding keys
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int t = t0 + sk + KPP * i;
        const bool in = d_ok && t < p.Tk;
        const int tt = in ? t : 0;
        const half8_t kv = *reinterpret_cast<const half8_t*>(Kb + (int64_t)tt * p.ldk);
        const half8_t vv = *reinterpret_cast<const half8_t*>(Vb + (int64_t)tt * p.ldv);
        kst[i] = in ? kv : zero8;
        vst[i] = in ? vv : zero8;
        if (ONES && sc == 7 && t < p.Tk) vst[i][7] = (half_t)1.0f;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int t = t0 + sk + KPP * i;
      const int d = sc * 8;
      // branch-free: every lane always loads 16 bytes from SOME valid address and the result is selected afterwards.
      // (Loads under divergent `if`s made hipcc wait for each one in turn: ~4 exposed HBM/L2 latencies per tile.)
      const bool in = (t < p.Tk) && (d < hd);
      const int64_t row = key_row(geo, b, in ? t : 0);
      const bool pad = row < 0;
      const int64_t hoff = (int64_t)h * hd + d;
      const half_t* kp = pad ? (padk ? padk + hoff : K) : K + (in ? row * p.ldk + hoff : 0);
      const half_t* vp = pad ? (padv ? padv + hoff : V) : V + (in ? row * p.ldv + hoff : 0);
      const half8_t kv = *reinterpret_cast<const half8_t*>(kp);
      const half8_t vv = *reinterpret_cast<const half8_t*>(vp);
      kst[i] = (in && (!pad || padk)) ? kv : zero8;
      vst[i] = (in && (!pad || padv)) ? vv : zero8;
      if (ONES && sc == 7 && t < p.Tk) vst[i][7] = (half_t)1.0f;  // column 63 of every real key: PV sums the probabilities
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int kk = sk + KPP * i;
      const int off = kk * RW + ((sc ^ (kk & (CPR - 1))) << 3);
      *reinterpret_cast<half8_t*>(&Ks[buf][off]) = kst[i];
      *reinterpret_cast<half8_t*>(&Vs[buf][off]) = vst[i];
    }
  };

  // exp(x*scale) = exp2(x*sl2); with REL the scores are first brought to natural units (s*scale + bias)
  const float sl2 = REL ? 1.44269504088896340736f : p.scale * 1.44269504088896340736f;
  const int ntile = (p.Tk + 63) / 64;
  // transposed-read addressing of the V fragment: lane (q4 = (lane&15)>>2, p4 = lane&3) of a 16-lane group points at
  // row key0+q4, columns 16*db + 4*p4 .. +3
  const int q4 = fr >> 2, p4 = fr & 3;
  // ---- LDS-DMA plan (DMA): wave w stages KB pieces 2w, 2w+1 of the K tile and of the V tile (8 rows x 128 B each);
  // the (chunk ^ row&7) swizzle goes on the per-lane SOURCE address, head-dim padding chunks and keys >= Tk fall outside
  // the descriptor (the row offset is in voffset, which the range check covers) and read as zeros.
  constexpr int LA = DMA ? NSL - 1 : 2, PT = 4;  // key tiles in flight ahead of the one being consumed
  __amdgpu_buffer_rsrc_t k_rs, v_rs;
  unsigned kvo[2], vvo[2];
  if constexpr (DMA) {
    const int64_t kb = ((int64_t)(p.Tk - 1) * p.ldk + hd) * 2, vb = ((int64_t)(p.Tk - 1) * p.ldv + hd) * 2;  // < 2^31: launcher
    k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(K + (int64_t)b * p.Tk * p.ldk + (int64_t)h * hd), 0, (int)kb, 0x00020000);
    v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(V + (int64_t)b * p.Tk * p.ldv + (int64_t)h * hd), 0, (int)vb, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ (row & 7);
      const bool ok = lc * 8 < hd;
      kvo[j] = ok ? (unsigned)(row * (int)p.ldk * 2 + lc * 16) : 0x80000000u;
      vvo[j] = ok ? (unsigned)(row * (int)p.ldv * 2 + lc * 16) : 0x80000000u;
    }
  }
  auto issue = [&](int it, int slot) {
    const unsigned ko = (unsigned)(it * 64 * (int)p.ldk * 2), vo = (unsigned)(it * 64 * (int)p.ldv * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      lds_dma16(k_rs, reinterpret_cast<char*>(&Ks[slot][0]) + (wave * 2 + j) * 1024, kvo[j] == 0x80000000u ? kvo[j] : kvo[j] + ko, 0);
      lds_dma16(v_rs, reinterpret_cast<char*>(&Vs[slot][0]) + (wave * 2 + j) * 1024, vvo[j] == 0x80000000u ? vvo[j] : vvo[j] + vo, 0);
    }
  };
  if constexpr (DMA) {
#pragma unroll
    for (int t = 0; t < LA; ++t)
      if (t < ntile) issue(t, t);
  } else {
    load_tile(0);
    store_tile(0);
    __syncthreads();
  }
  // Loop-invariant lane parts of the LDS fragment addresses (in halfs), so that a tile pays one add per base for the ring
  // slot and the rest are instruction-offset immediates (hipcc otherwise re-derives ~45 integer VALU per tile from `buf`):
  // K fragment of k-step ks, key block kb: k_lane[ks] + kb * 16 * RW;  V^T fragment of d-block db, k-step ks: v_lane[db] + ks * 32 * RW (+ 16 * RW)
  int k_lane[KS], v_lane[NDB];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_lane[ks] = fr * RW + ((((ks << 2) + fg) ^ (fr & (CPR - 1))) << 3);
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    const int chunk = db * 2 + (p4 >> 1), rl = fg * 4 + q4;  // (rl + 32 ks (+ 16)) & (CPR - 1) == rl & (CPR - 1): CPR <= 16
    v_lane[db] = rl * RW + ((chunk ^ (rl & (CPR - 1))) << 3) + (p4 & 1) * 4;
  }
  // The body is instantiated twice: full tiles carry NO masking code at all (hipcc otherwise if-converts the
  // wave-uniform `partial` test into a compare+select per score element: ~60 VALU per tile), the last tile masks.
  auto tile_body = [&](const int it, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int t0 = it * 64;
    int buf;
    if constexpr (DMA) {
      // tile `it` has landed once only this wave's DMAs of the next tile are outstanding (Q's loads are older; nothing else
      // is loaded from global memory in the loop)
      // (a deeper ring does not help: DMA latency is not what the waves wait for — 4 / 5 slots measured 10-20 % slower)
      const int left = ntile - 1 - it;
      wait_tiles<PT>(left < LA - 1 ? left : LA - 1);
      __builtin_amdgcn_s_barrier();
      if (it + LA < ntile) issue(it + LA, (it + LA) % NSL);
      buf = it % NSL;
    } else {
      buf = it & 1;
      if (it + 1 < ntile) load_tile(t0 + 64);
    }

    // one base per (k-step | d-block) and ring slot; everything else below is an instruction-offset immediate
    unsigned kbase[KS], vbase[NDB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kbase[ks] = lds_addr(&Ks[buf][k_lane[ks]]);
#pragma unroll
    for (int db = 0; db < NDB; ++db) vbase[db] = lds_addr(&Vs[buf][v_lane[db]]);

    // ---- S^T = K . Q^T  (raw, unscaled)
    f32x4 sacc[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) sacc[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const half8_t kf = lds_read8(kbase[ks], kb * 16 * RW * 2);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          sacc[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qb][ks], sacc[qb][kb], 0, 0, 0);
      }
    }

    // ---- online softmax (per query = per lane column); P^T fragments come straight from the accumulators
    half8_t pf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (REL) {
        if (p.rel_S == 64) {
          // global attention of the 64 x 64 grid: a 64-key tile is exactly grid row `it`, so the row term is one value per
          // query and tile and the column terms of a lane's four keys are one 8-byte load — no index arithmetic
          const float by = (float)relq[qb][it];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const half4_t bx = *reinterpret_cast<const half4_t*>(relq[qb] + 64 + kb * 16 + fg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[qb][kb][r] = fmaf(sacc[qb][kb][r], p.scale, by + (float)bx[r]);
          }
        } else {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int key = t0 + kb * 16 + fg * 4 + r;
              key = key < p.Tk ? key : 0;
              const int ky = key / p.rel_S, kx = key - ky * p.rel_S;
              const float bias = (float)relq[qb][ky] + (float)relq[qb][p.rel_S + kx];
              sacc[qb][kb][r] = fmaf(sacc[qb][kb][r], p.scale, bias);
            }
        }
      }
      if (MASK) {  // only the last tile can hold keys >= Tk
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t0 + kb * 16 + fg * 4 + r >= p.Tk) sacc[qb][kb][r] = -INFINITY;
      }
      float mx = fmaxf(fmaxf(sacc[qb][0][0], sacc[qb][0][1]), fmaxf(sacc[qb][0][2], sacc[qb][0][3]));
#pragma unroll
      for (int kb = 1; kb < 4; ++kb)
        mx = fmaxf(mx, fmaxf(fmaxf(sacc[qb][kb][0], sacc[qb][kb][1]), fmaxf(sacc[qb][kb][2], sacc[qb][kb][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx);
      // the running maximum settles after the first few key tiles: rescale the accumulators only when some query of
      // the wave raised it (wave-uniform branch) — saves an exp2 and 16 multiplies per query block and tile
      // (flat geometry only: measured slower on the windowed shapes, whose four key tiles mostly do raise the maximum)
      const bool grew = nQT < 0 || __builtin_amdgcn_ballot_w64(m_new > m_run[qb]) != 0;
      const float alpha = grew ? __builtin_amdgcn_exp2f((m_run[qb] - m_new) * sl2) : 1.0f;
      m_run[qb] = m_new;
      const float mb = m_new * sl2;
      float rs = 0.f;
      const half2_t ones2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r], sl2, -mb));
          const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r + 1], sl2, -mb));
          const half2_t e = {(half_t)e0, (half_t)e1};
          if (!ONES) rs = __builtin_amdgcn_fdot2(e, ones2, rs, false);  // sums exactly what the PV MFMA sees
          pf[qb][kb >> 1][(kb & 1) * 4 + r] = e[0];
          pf[qb][kb >> 1][(kb & 1) * 4 + r + 1] = e[1];
        }
      if (!ONES) l_run[qb] = l_run[qb] * alpha + rs;
      if (grew) {
#pragma unroll
        for (int db = 0; db < NDB; ++db) oacc[qb][db] *= alpha;
      }
    }

    // ---- O^T += V^T . P^T   (V^T fragments by hardware-transposed LDS reads of the row-major V tile)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const half4_t lo = lds_tr_read_at(vbase[db], ks * 32 * RW * 2);
        const half4_t hi = lds_tr_read_at(vbase[db], (ks * 32 + 16) * RW * 2);
        half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (DMA && ONES && db == 3) {  // row d = 63 of V^T (lanes fr == 15) is the ones row: the staged tile holds zeros there
          const half_t one = (half_t)1.0f;
          const half8_t ones8 = {one, one, one, one, one, one, one, one};
          vf = fr == 15 ? ones8 : vf;
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qb][ks], oacc[qb][db], 0, 0, 0);
      }
    }

    if constexpr (!DMA) {
      if (it + 1 < ntile) store_tile(buf ^ 1);
      __syncthreads();
    }
  };
  for (int it = 0; it + 1 < ntile; ++it) tile_body(it, std::false_type{});
  if (p.Tk & 63)
    tile_body(ntile - 1, std::true_type{});
  else
    tile_body(ntile - 1, std::false_type{});

  // ---- epilogue: lane owns O[q = fr][d = 16*db + 4*fg + r]
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l;
    if (ONES) {
      l = __shfl(oacc[qb][3][3], 48 + fr, 64);  // O[q][63] lives in lane group 3, register 3 of the last d-block
    } else {
      l = l_run[qb];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    if (qrow[qb] < 0) continue;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      const int d = db * 16 + fg * 4;
      if (d >= hd) continue;
      half4_t o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (half_t)(oacc[qb][db][r] * inv);
      *reinterpret_cast<half4_t*>(O + qrow[qb] * p.ldo + (int64_t)h * hd + d) = o;
    }
  }
}


// ---- long flat sequences, software-pipelined (round 3; Hiera's global blocks: T = 4096, head dim 56).  attn_kernel's LDS-DMA
// form issues, per 64-key tile and wave, 16 S MFMAs, then ~180 VALU instructions of softmax that depend on them, then 16 PV MFMAs
// that depend on those, and — its LDS images being static __shared__ arrays — hipcc puts `s_waitcnt vmcnt(0)` in front of the
// first fragment read after every DMA issue, so its ring never has a tile in flight.  Here:
//  * the images are DYNAMIC LDS and V's transposed reads are inline asm (the builtin form gets the same conservative wait), so the
//    counted waits are the only ones and two groups of DMAs stay in flight;
//  * the K ring runs ONE TILE AHEAD of the V ring and the wave issues S(i+1) before the softmax of tile i: the next tile's MFMAs
//    run under this tile's exponentials inside the same wave (a second accumulator set, 32 registers; sched_group_barrier
//    interleaves one MFMA per four VALU instructions);
//  * the one wave-uniform branch of a tile (did a running maximum grow? then rescale) sits before that block;
//  * the row maximum is v_med3_f32 against +inf (fmaxf on MFMA results costs hipcc a canonicalising v_max per operand) and its cross-lane step
//    v_permlane16_swap / v_permlane32_swap instead of two ds_bpermute round trips in the middle of the dependency chain.
// Same MFMA order per accumulator, same exp2 arguments, same f16 roundings as attn_kernel<2, ONES, false, true, 64>: identical
// bits (tools/attn_gp_probe.py), 1601 -> 1518 us at B = 30, H = 8 (profiles/r03_attn_gp.txt).  What the time is made of
// (-DLMX_GP_DBG=n builds, tools/attn_gp_decompose.sh): without the exponentials -10 %, without the row maximum -10 %, without the
// DMA issue -15 %, without either MFMA group and its fragment reads -32 % each: the phases add up — per wave and tile ~1200 issue
// cycles (32 MFMAs hold the vector issue for 8 of their 16 cycles, 34 v_exp_f32 at 8, ~150 other VALU at 4, 4 DMA pieces at
// 60 - 100) against 512 matrix cycles.  At head dim 64 a score costs the same softmax as at 128 and feeds half the MFMA work.
// (v_med3_f32 against +inf rather than inline-asm v_max3_f32: the operands are MFMA results, and hipcc inserts the wait states an
// MFMA write -> VALU read needs only in front of instructions it knows; csrc/hiera.hip has the case that showed it)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(a, b, INFINITY), c, INFINITY);
}
// maximum over the four 16-lane rows of a wave, delivered to every row (lane maps: tools/permlane_swap_probe.hip).  Two traps:
// __builtin_bit_cast(float, a[1]) on an element of the returned 2-vector reads element 0 (hipcc 7.2 takes the vector's address):
// the maximum then covered one row only — still a valid softmax shift, so only the bit-for-bit comparison with the shuffle form
// showed it — hence the scalar temporaries; and the instructions next to a swap stay the compiler's own (fmaxf, not the asm
// helpers above), because gfx950 needs wait states around v_permlane*_swap that hipcc inserts only for instructions it knows.
__device__ __forceinline__ float row_max4(float ma, float mc) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned u = __builtin_bit_cast(unsigned, fmaxf(ma, mc));
  const u32x2 a = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // {r0 r0 r2 r2}, {r1 r1 r3 r3}
  const unsigned a0 = a[0], a1 = a[1];
  const unsigned u1 = __builtin_bit_cast(unsigned, fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1)));
  const u32x2 b = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);  // {lo lo}, {hi hi}
  const unsigned b0 = b[0], b1 = b[1];
  return fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
}

#ifndef LMX_GP_DBG
#define LMX_GP_DBG 0
#endif
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// NWV waves share a K / V tile (64 * NWV / 2 ... 32 queries per wave): with 8 waves a wave issues ONE LDS-DMA piece per tensor and tile
// instead of two (a piece costs the issuing wave 60 - 100 cycles beside MFMAs), at the price of a barrier among eight waves
template <bool ONES, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void attn_gp_kernel(const lmx_attn_desc p, const int nQT) {
  constexpr int QB = 2, NSL = 3, RW = 64, NJ = 8 / NWV, PT = 2 * NJ;  // NJ pieces per wave, tensor and tile (16 pieces of 8 rows per tile)
  // DYNAMIC LDS on purpose: for a static __shared__ array hipcc's waitcnt pass knows the LDS-DMA writes and the fragment reads
  // touch the same object and puts `s_waitcnt vmcnt(0)` in front of the first read after every issue — the ring then never has a
  // tile in flight (attn_kernel's DMA form has exactly that wait in its loop).  The counted waits below are the synchronisation.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t (*Ks)[64 * RW] = reinterpret_cast<half_t (*)[64 * RW]>(smem);
  half_t (*Vs)[64 * RW] = reinterpret_cast<half_t (*)[64 * RW]>(smem + NSL * 64 * RW * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int bid;
  {
    const int n = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = n >> 3, r = n & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nQTa = nQT < 0 ? -nQT : nQT;
  const int qt = bid % nQTa;
  bid /= nQTa;
  const int h = bid % p.H;
  const int b = bid / p.H;
  const half_t* Q = reinterpret_cast<const half_t*>(p.Q);
  const half_t* K = reinterpret_cast<const half_t*>(p.K);
  const half_t* V = reinterpret_cast<const half_t*>(p.V);
  half_t* O = reinterpret_cast<half_t*>(p.O);
  const int hd = p.hd;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  const int q_base = qt * (NWV * 32) + wave * 32;
  half8_t qf[QB][2];
  int64_t qrow[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = q_base + qb * 16 + fr;
    qrow[qb] = (tq < p.Tq) ? (int64_t)b * p.Tq + tq : -1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int d = ks * 32 + fg * 8;
      const bool ok = qrow[qb] >= 0 && d < hd;
      const half8_t qv = *reinterpret_cast<const half8_t*>(Q + (ok ? qrow[qb] * p.ldq + (int64_t)h * hd + d : 0));
      qf[qb][ks] = ok ? qv : zero8;
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
  const float sl2 = p.scale * 1.44269504088896340736f;
  const int ntile = (p.Tk + 63) / 64;
  const int q4 = fr >> 2, p4 = fr & 3;

  // LDS-DMA plan as in attn_kernel: wave w stages pieces 2w, 2w+1 (8 rows x 128 B each) of a K tile and of a V tile
  const int64_t kbytes = ((int64_t)(p.Tk - 1) * p.ldk + hd) * 2, vbytes = ((int64_t)(p.Tk - 1) * p.ldv + hd) * 2;  // < 2^31: launcher
  const __amdgpu_buffer_rsrc_t k_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(K + (int64_t)b * p.Tk * p.ldk + (int64_t)h * hd), 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(V + (int64_t)b * p.Tk * p.ldv + (int64_t)h * hd), 0, (int)vbytes, 0x00020000);
  unsigned kvo[NJ], vvo[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = (wave * NJ + j) * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ (row & 7);
    const bool ok = lc * 8 < hd;
    kvo[j] = ok ? (unsigned)(row * (int)p.ldk * 2 + lc * 16) : 0x80000000u;
    vvo[j] = ok ? (unsigned)(row * (int)p.ldv * 2 + lc * 16) : 0x80000000u;
  }
  // a tile index past the last one addresses rows >= Tk: outside the descriptor, the DMA writes zeros (into a free slot)
  auto issue_k = [&](int it) {
    const unsigned ko = (unsigned)(it * 64 * (int)p.ldk * 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      lds_dma16(k_rs, reinterpret_cast<char*>(&Ks[it % NSL][0]) + (wave * NJ + j) * 1024, kvo[j] == 0x80000000u ? kvo[j] : kvo[j] + ko, 0);
  };
  auto issue_v = [&](int it) {
    const unsigned vo = (unsigned)(it * 64 * (int)p.ldv * 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      lds_dma16(v_rs, reinterpret_cast<char*>(&Vs[it % NSL][0]) + (wave * NJ + j) * 1024, vvo[j] == 0x80000000u ? vvo[j] : vvo[j] + vo, 0);
  };
  int k_lane[2], v_lane[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) k_lane[ks] = fr * RW + ((((ks << 2) + fg) ^ (fr & 7)) << 3);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int chunk = db * 2 + (p4 >> 1), rl = fg * 4 + q4;
    v_lane[db] = rl * RW + ((chunk ^ (rl & 7)) << 3) + (p4 & 1) * 4;
  }
  auto s_mfma = [&](int it, f32x4 (&s)[QB][4]) {  // S^T(it) = K(it) . Q^T, raw
    unsigned kbase[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kbase[ks] = lds_addr(&Ks[it % NSL][k_lane[ks]]);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) s[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const half8_t kf = lds_read8(kbase[ks], kb * 16 * RW * 2);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qb][ks], s[qb][kb], 0, 0, 0);
      }
  };

  // group g of DMAs = { K tile g+1, V tile g }: 4 per wave; K tile 0 goes first on its own
  issue_k(0);
  issue_k(1);
  issue_v(0);
  if (ntile > 1) {
    issue_k(2);
    issue_v(1);
    wait_vmcnt<2 * PT>();
  } else {
    wait_vmcnt<PT>();
  }
  __builtin_amdgcn_s_barrier();
  f32x4 sA[QB][4], sB[QB][4];
  s_mfma(0, sA);

  // one key tile: `cur` holds S(it); S(it+1) goes to `nxt`
  auto step = [&](const int it, f32x4 (&cur)[QB][4], f32x4 (&nxt)[QB][4], auto mask_tag, auto next_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    constexpr bool NEXT = decltype(next_tag)::value;  // a tile it+1 exists
    const int t0 = it * 64;
    const int left = ntile - 1 - it;
    wait_tiles<PT>(left < 1 ? left : 1);  // groups <= it have landed: K(it+1), V(it)
    __builtin_amdgcn_s_barrier();         // ... for every wave, and every wave is through S(it) and PV(it-1)
#if LMX_GP_DBG == 5  // decomposition build: no LDS-DMA in the loop (the waits fall through)
    if (p.Tk < 0)
#endif
    if (it + 2 < ntile) {
      issue_k(it + 3);  // slot of K(it)
      issue_v(it + 2);  // slot of V(it-1)
    }
    // ---- row maxima of S(it) first: the one wave-uniform branch of a tile (did any query's maximum grow?) comes BEFORE the block
    // that holds the next tile's S MFMAs and this tile's exponentials, so the scheduler can interleave those
    float nmb[QB], alq[QB];
    {
      float alpha[QB];
      bool grew = nQT < 0;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        if (MASK) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (t0 + kb * 16 + fg * 4 + r >= p.Tk) cur[qb][kb][r] = -INFINITY;
        }
#if LMX_GP_DBG == 4  // decomposition build: no maximum
        const float m_new = 0.f;
#else
        float ma = vmax3(cur[qb][0][0], cur[qb][0][1], cur[qb][0][2]);
        float mc = vmax3(cur[qb][2][0], cur[qb][2][1], cur[qb][2][2]);
        ma = vmax3(ma, cur[qb][0][3], cur[qb][1][0]);
        mc = vmax3(mc, cur[qb][2][3], cur[qb][3][0]);
        ma = vmax3(ma, cur[qb][1][1], cur[qb][1][2]);
        mc = vmax3(mc, cur[qb][3][1], cur[qb][3][2]);
        ma = vmax3(ma, cur[qb][1][3], cur[qb][3][3]);
        const float m_new = fmaxf(m_run[qb], row_max4(ma, mc));
#endif
        grew = grew || __builtin_amdgcn_ballot_w64(m_new > m_run[qb]) != 0;
        alpha[qb] = (m_run[qb] - m_new) * sl2;
        alq[qb] = 1.0f;
        m_run[qb] = m_new;
        nmb[qb] = -(m_new * sl2);
      }
      // the running maximum settles after the first few key tiles: rescale only when some query of the wave raised it
      // (exp2(0) = 1 for the queries that did not: their accumulators are multiplied by exactly 1)
      if (grew) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float al = __builtin_amdgcn_exp2f(alpha[qb]);
          alq[qb] = al;
#pragma unroll
          for (int db = 0; db < 4; ++db) oacc[qb][db] *= al;
        }
      }
    }
#if LMX_GP_DBG == 2  // decomposition build: no S MFMAs (and no K fragment reads) in the loop
    if (p.Tk < 0) s_mfma(it + 1, nxt);
#else
    if constexpr (NEXT) s_mfma(it + 1, nxt);
#endif
    half8_t pf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float rs = 0.f;
      const half2_t ones2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 a = {fmaf(cur[qb][kb][r], sl2, nmb[qb]), fmaf(cur[qb][kb][r + 1], sl2, nmb[qb])};
#if LMX_GP_DBG == 1  // decomposition build: no transcendental
          const float e0 = a[0], e1 = a[1];
#else
          const float e0 = __builtin_amdgcn_exp2f(a[0]);
          const float e1 = __builtin_amdgcn_exp2f(a[1]);
#endif
          const half2_t e = {(half_t)e0, (half_t)e1};
          if (!ONES) rs = __builtin_amdgcn_fdot2(e, ones2, rs, false);
          pf[qb][kb >> 1][(kb & 1) * 4 + r] = e[0];
          pf[qb][kb >> 1][(kb & 1) * 4 + r + 1] = e[1];
        }
      if (!ONES) l_run[qb] = l_run[qb] * alq[qb] + rs;
    }
#ifndef LMX_ATTN_GP_NOSCHED
    if constexpr (NEXT) {  // one S MFMA (16 matrix cycles) per four VALU instructions of the exponentials
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    }
#endif
    // ---- O^T += V^T . P^T.  The transposed reads are inline asm: through the builtin hipcc cannot tell them from the LDS-DMA
    // writes in flight and waits for vmcnt(0) first.  All sixteen are requested, then waited for together.
#if LMX_GP_DBG == 3  // decomposition build: no PV MFMAs (and no V fragment reads)
    if (p.Tk < 0)
#endif
    {
    unsigned vbase[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vbase[db] = lds_addr(&Vs[it % NSL][v_lane[db]]);
    u32x2_t vt[2][4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[0][db][0]) : "v"(vbase[db]) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vt[0][db][1]) : "v"(vbase[db]) : "memory");
    }
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(vt[1][db][0]) : "v"(vbase[db]) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(vt[1][db][1]) : "v"(vbase[db]) : "memory");
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // (the registers are operands of the wait so that no consumer can be scheduled in front of it)
#define LMX_VT8(k) "+v"(vt[k][0][0]), "+v"(vt[k][0][1]), "+v"(vt[k][1][0]), "+v"(vt[k][1][1]), "+v"(vt[k][2][0]), "+v"(vt[k][2][1]), "+v"(vt[k][3][0]), "+v"(vt[k][3][1])
      if (ks == 0)
        asm volatile("s_waitcnt lgkmcnt(8)" : LMX_VT8(0)::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : LMX_VT8(1)::"memory");
#undef LMX_VT8
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        u32x4_t w = {vt[ks][db][0][0], vt[ks][db][0][1], vt[ks][db][1][0], vt[ks][db][1][1]};
        if (ONES && db == 3) {  // row d = 63 of V^T (lanes fr == 15) is the ones row: the staged tile holds zeros there
          const u32x4_t ones4 = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
          w = fr == 15 ? ones4 : w;
        }
        const half8_t vf = __builtin_bit_cast(half8_t, w);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qb][ks], oacc[qb][db], 0, 0, 0);
      }
    }
    }
#if LMX_GP_DBG == 3
    for (int qb = 0; qb < QB; ++qb) oacc[qb][0][0] += (float)pf[qb][0][0] + (float)pf[qb][1][7] + (float)pf[qb][0][3] + (float)pf[qb][1][4];
#endif
  };
  const std::true_type T{};
  const std::false_type F{};
  int it = 0;
  for (; it + 2 < ntile; it += 2) {
    step(it, sA, sB, F, T);
    step(it + 1, sB, sA, F, T);
  }
  const bool part = (p.Tk & 63) != 0;
  if (it + 2 == ntile) {
    step(it, sA, sB, F, T);
    if (part)
      step(it + 1, sB, sA, T, F);
    else
      step(it + 1, sB, sA, F, F);
  } else {
    if (part)
      step(it, sA, sB, T, F);
    else
      step(it, sA, sB, F, F);
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l;
    if (ONES) {
      l = __shfl(oacc[qb][3][3], 48 + fr, 64);
    } else {
      l = l_run[qb];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
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


// ---- tiny sequences (Tq <= 16 and Tk <= 16: Hiera's 4 x 4 windows, 16 384 of them per image batch and head).
// attn_kernel spends a 4-wave workgroup, a 64 x 64 LDS tile and a barrier on each (window, head) and uses 1/16 of its
// MFMA work; here ONE WAVE owns an item and nothing is shared: Q and K fragments come straight from global memory (a lane's
// 16 bytes are contiguous in both), V goes through a wave-private 2 KB LDS slice only to be read back transposed, one
// 16 x 16 x 32 MFMA pair gives S^T, one per 16-wide d block gives O^T.  No barrier, 8 KB of LDS per workgroup.
__global__ __launch_bounds__(256) void attn_small_kernel(const lmx_attn_desc p, const Geo geo, const int items) {
  __shared__ __attribute__((aligned(16))) half_t Vs[4][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;  // whole wave (items are per wave; there is no barrier in this kernel)
  const int h = item % p.H, b = item / p.H;
  const half_t* Q = reinterpret_cast<const half_t*>(p.Q);
  const half_t* K = reinterpret_cast<const half_t*>(p.K);
  const half_t* V = reinterpret_cast<const half_t*>(p.V);
  const half_t* padk = reinterpret_cast<const half_t*>(p.pad_k);
  const half_t* padv = reinterpret_cast<const half_t*>(p.pad_v);
  const int hd = p.hd;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // V: lane -> key row lane>>2, the two 16-byte chunks 2*(lane&3), +1 (branch-free loads, selected afterwards)
  {
    const int r = lane >> 2, c0 = (lane & 3) * 2;
    const bool in = r < p.Tk;
    const int64_t row = key_row(geo, b, in ? r : 0);
    const bool pad = row < 0;
    half_t* dst = &Vs[wave][r * 64];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = c0 + e, d = c * 8;
      const bool ok = in && d < hd && (!pad || padv);
      const half_t* vp = (in && d < hd) ? (pad ? (padv ? padv + (int64_t)h * hd + d : V) : V + row * p.ldv + (int64_t)h * hd + d) : V;
      const half8_t vv = *reinterpret_cast<const half8_t*>(vp);
      *reinterpret_cast<half8_t*>(dst + ((c ^ (r & 7)) << 3)) = ok ? vv : zero8;
    }
  }
  // Q (B operand) and K (A operand) fragments: lane (fr, fg) holds row fr, features 32*ks + 8*fg .. +7
  const int64_t qrow = fr < p.Tq ? query_row(geo, b, fr) : -1;
  const bool kin = fr < p.Tk;
  const int64_t krow = key_row(geo, b, kin ? fr : 0);
  const bool kpad = krow < 0;
  half8_t qf[2], kf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int d = ks * 32 + fg * 8;
    const bool dq = qrow >= 0 && d < hd;
    const half8_t qv = *reinterpret_cast<const half8_t*>(Q + (dq ? qrow * p.ldq + (int64_t)h * hd + d : 0));
    qf[ks] = dq ? qv : zero8;
    const bool dk = kin && d < hd;
    const half_t* kp = dk ? (kpad ? (padk ? padk + (int64_t)h * hd + d : K) : K + krow * p.ldk + (int64_t)h * hd + d) : K;
    const half8_t kv = *reinterpret_cast<const half8_t*>(kp);
    kf[ks] = (dk && (!kpad || padk)) ? kv : zero8;
  }
  // S^T[key][query]: lane (fr = query, fg) holds keys 4*fg + i
  f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
  sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[0], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[1], sacc, 0, 0, 0);
  const float sl2 = p.scale * 1.44269504088896340736f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (fg * 4 + i >= p.Tk) sacc[i] = -INFINITY;
  float mx = fmaxf(fmaxf(sacc[0], sacc[1]), fmaxf(sacc[2], sacc[3]));
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float mb = mx * sl2;
  half8_t pf = zero8;  // k-slot 8*fg + j <-> key 4*fg + j for j < 4 (attn_kernel's permutation at ks = 0); slots j >= 4 are keys >= 16
  float l = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const half_t e = (half_t)__builtin_amdgcn_exp2f(fmaf(sacc[i], sl2, -mb));
    pf[i] = e;
    l += (float)e;  // sums exactly what the PV MFMA sees
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  // O^T = V^T . P^T: V^T fragments by transposed reads of the wave's own slice (same addressing as attn_kernel, key rows 0..15)
  const int q4 = fr >> 2, p4 = fr & 3;
  const int r0 = fg * 4 + q4;
  half_t* O = reinterpret_cast<half_t*>(p.O);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int chunk = db * 2 + (p4 >> 1);
    const half4_t lo = lds_tr_read(&Vs[wave][r0 * 64 + ((chunk ^ (r0 & 7)) << 3) + (p4 & 1) * 4]);
    const half8_t vf = {lo[0], lo[1], lo[2], lo[3], 0, 0, 0, 0};
    f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
    oacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, oacc, 0, 0, 0);
    const int d = db * 16 + fg * 4;
    if (qrow >= 0 && d < hd) {
      const half4_t o = {(half_t)(oacc[0] * inv), (half_t)(oacc[1] * inv), (half_t)(oacc[2] * inv), (half_t)(oacc[3] * inv)};
      *reinterpret_cast<half4_t*>(O + qrow * p.ldo + (int64_t)h * hd + d) = o;
    }
  }
}

// Decomposed relative-position tables on MFMA (TF:models/sam/modeling_sam.py get_decomposed_rel_pos / add_decomposed_rel_pos:
// rel_h[b,h,(ty,tx),j] = q . Rh[ty - j + S - 1],  rel_w[b,h,(ty,tx),j] = q . Rw[tx - j + S - 1]).
// For a fixed ty the S tokens of a grid row share their S rows of Rh, for a fixed tx the S tokens of a grid column share their
// rows of Rw: each is a [S x hd] x [hd x S] product.  One wave = one (b, h, axis, line): A operand = the S relative-position
// rows (f32 table -> f16 on the fly), B operand = the line's query rows (f16, straight from global memory), accumulator
// lane (token, 4 consecutive j) -> one 8-byte store into the token's table row.  (The first version was one thread per
// token looping over 2S x hd products: 1.6 ms per layer at S = 64, more than the layer's GEMMs.)
template <int KS>
__global__ __launch_bounds__(256) void relpos_tables_kernel(const lmx_attn_desc p, const Geo geo, const float* __restrict__ rh,
                                                            const float* __restrict__ rw, int S, half_t* __restrict__ out,
                                                            const int64_t items) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= items) return;  // whole wave; no barrier in this kernel
  // item -> (b, h, axis, line)
  const int line = (int)(item % S);
  const int axis = (int)((item / S) & 1);  // 0: rows share Rh (line = ty), 1: columns share Rw (line = tx)
  const int64_t bh = item / (2 * S);
  const int h = (int)(bh % p.H), b = (int)(bh / p.H);
  const int hd = p.hd;
  const float* R = axis ? rw : rh;
  const half_t* Q = reinterpret_cast<const half_t*>(p.Q);
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const int nblk = (S + 15) / 16;
  for (int mb = 0; mb < nblk; ++mb) {  // 16 tokens of the line
    const int pos = mb * 16 + fr;      // position along the line
    const int t = axis ? pos * S + line : line * S + pos;
    const int64_t qrow = pos < S ? query_row(geo, b, t) : -1;
    half8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d = ks * 32 + fg * 8;
      const bool ok = qrow >= 0 && d < hd;
      const half8_t qv = *reinterpret_cast<const half8_t*>(Q + (ok ? qrow * p.ldq + (int64_t)h * hd + d : 0));
      qf[ks] = ok ? qv : zero8;
    }
    half_t* orow = out + ((bh * p.Tq) + (pos < S ? t : 0)) * (2 * S) + axis * S;
    for (int jb = 0; jb < nblk; ++jb) {  // 16 relative offsets
      const int j = jb * 16 + fr;        // A-operand row of this lane
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d = ks * 32 + fg * 8;
        const bool ok = j < S && d < hd;
        const float* rp = R + (ok ? (int64_t)(line - j + S - 1) * hd + d : 0);
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + (ok ? 4 : 0));
        const half8_t rf = {(half_t)r0[0], (half_t)r0[1], (half_t)r0[2], (half_t)r0[3],
                            (half_t)r1[0], (half_t)r1[1], (half_t)r1[2], (half_t)r1[3]};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ok ? rf : zero8, qf[ks], acc, 0, 0, 0);
      }
      // acc: lane (fr = token, fg) holds offsets j = 16*jb + 4*fg + i
      const int j0 = jb * 16 + fg * 4;
      if (pos < S && j0 < S) {
        if (j0 + 3 < S && (S & 3) == 0) {
          const half4_t o = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
          *reinterpret_cast<half4_t*>(orow + j0) = o;
        } else {
          for (int i = 0; i < 4 && j0 + i < S; ++i) orow[j0 + i] = (half_t)acc[i];
        }
      }
    }
  }
}


// ---- whole-sequence tiles: 128 < Tk <= 208 keys and Tq <= 208 queries, head dim <= 64 (Hiera-B+ stage 3's 14 x 14 windows:
// 196 tokens, 400 windows x 8 heads per 16-frame pass; DINOv3's 201 tokens).  attn_kernel walks these as 64-key tiles
// with an online softmax and 128-query workgroups: 196 keys cost four tiles (the fourth holds 4 keys), 196 queries cost two
// workgroups (the second holds 68) that each stage all of K and V — 1.7x the MFMA and softmax work and 2x the staging of
// what the problem holds.  Here ONE workgroup owns a (window | image, head): K (13 blocks of 16 keys) and V are staged
// once, row-major and swizzled exactly as in attn_kernel, and a wave takes the 16-query blocks two at a time (each K / V^T
// fragment read from LDS feeds two MFMAs): S^T for ALL keys sits in the accumulators (2 x 13 x f32x4), so the softmax is
// a plain one - max, exp2, (ones column | dot2) sum - with no running maximum and no rescale; then O^T += V^T . P^T over
// seven 32-key k-steps.  55 KB of LDS and < 256 VGPRs: two workgroups per CU, one staging while the other computes.
template <int QB, bool ONES>
__global__ __launch_bounds__(256, QB == 2 ? 2 : 3) void attn_sp_kernel(const lmx_attn_desc p, const Geo geo) {
  constexpr int RW = 64, CPR = 8, KB = 13, KROWS = KB * 16, VROWS = KROWS, NPASS = (KROWS + 31) / 32;
  __shared__ __attribute__((aligned(16))) half_t Ks[KROWS * RW];
  __shared__ __attribute__((aligned(16))) half_t Vs[VROWS * RW];  // rows Tk..207 are zeros; 2 x 26 KB: three workgroups fit a CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int bid;
  {  // XCD-aware bijective remap: the heads of one window read the same cache lines (a token row holds all heads)
    const int n = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = n >> 3, r = n & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int h = bid % p.H, b = bid / p.H;
  const half_t* Q = reinterpret_cast<const half_t*>(p.Q);
  const half_t* K = reinterpret_cast<const half_t*>(p.K);
  const half_t* V = reinterpret_cast<const half_t*>(p.V);
  const half_t* padk = reinterpret_cast<const half_t*>(p.pad_k);
  const half_t* padv = reinterpret_cast<const half_t*>(p.pad_v);
  half_t* O = reinterpret_cast<half_t*>(p.O);
  const int hd = p.hd;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- stage K and V: chunk sc of key sk + 32 i (branch-free loads, selected afterwards, as attn_kernel's load_tile)
  {
    const int sc = tid & (CPR - 1), sk = tid / CPR;
    const int d = sc * 8;
    const int64_t hoff = (int64_t)h * hd + d;
    // two rounds (passes 0-3, then 4-6): 16 + 12 loads in flight per thread instead of 28 keeps the register peak of
    // the staging below that of the compute phase; the other workgroups of the CU cover the second round trip
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int H0 = 4;
      half8_t kst[H0], vst[H0];
#pragma unroll
      for (int j = 0; j < H0; ++j) {
        const int i = half * H0 + j;
        if (i >= NPASS) continue;
        const int t = sk + 32 * i;
        const bool in = (t < p.Tk) && (d < hd);
        const int64_t row = geo.mode == 0 ? (int64_t)b * p.Tk + (in ? t : 0) : key_row(geo, b, in ? t : 0);
        const bool pad = row < 0;
        const half_t* kp = pad ? (padk ? padk + hoff : K) : K + (in ? row * p.ldk + hoff : 0);
        const half_t* vp = pad ? (padv ? padv + hoff : V) : V + (in ? row * p.ldv + hoff : 0);
        const half8_t kv = *reinterpret_cast<const half8_t*>(kp);
        const half8_t vv = *reinterpret_cast<const half8_t*>(vp);
        kst[j] = (in && (!pad || padk)) ? kv : zero8;
        vst[j] = (in && (!pad || padv)) ? vv : zero8;
        if (ONES && sc == 7 && t < p.Tk) vst[j][7] = (half_t)1.0f;  // column 63 of every real key: PV sums the probabilities
      }
#pragma unroll
      for (int j = 0; j < H0; ++j) {
        const int i = half * H0 + j;
        if (i >= NPASS) continue;
        const int kk = sk + 32 * i;
        const int off = kk * RW + ((sc ^ (kk & (CPR - 1))) << 3);
        if (kk < KROWS) {
          *reinterpret_cast<half8_t*>(&Ks[off]) = kst[j];
          *reinterpret_cast<half8_t*>(&Vs[off]) = vst[j];
        }
      }
    }
  }
  __syncthreads();

  const float sl2 = p.scale * 1.44269504088896340736f;
  const int q4 = fr >> 2, p4 = fr & 3;
  const int nqb = (p.Tq + 15) / 16, npair = (nqb + QB - 1) / QB;
  for (int pr = wave; pr < npair; pr += 4) {  // no barrier below: waves run their pairs independently
    // ---- Q fragments (B operand) of the two 16-query blocks
    half8_t qf[QB][2];
    int64_t qrow[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int tq = (pr * QB + qb) * 16 + fr;
      qrow[qb] = (tq < p.Tq) ? query_row(geo, b, tq) : -1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int d = ks * 32 + fg * 8;
        const bool ok = qrow[qb] >= 0 && d < hd;
        const half8_t qv = *reinterpret_cast<const half8_t*>(Q + (ok ? qrow[qb] * p.ldq + (int64_t)h * hd + d : 0));
        qf[qb][ks] = ok ? qv : zero8;
      }
    }
    // ---- S^T = K . Q^T for all 13 key blocks
    f32x4 sacc[QB][KB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) sacc[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = (((ks << 2) + fg) ^ (fr & (CPR - 1))) << 3;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(&Ks[(kb * 16 + fr) * RW + coff]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qb][ks], sacc[qb][kb], 0, 0, 0);
      }
    }
    // ---- softmax over the whole key range (per query = per lane column), P^T fragments straight from the accumulators
    half8_t pf[QB][7];
    float l_sum[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int kb = 8; kb < KB; ++kb)  // Tk > 128: only blocks 8.. can hold keys >= Tk
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb * 16 + fg * 4 + r >= p.Tk) sacc[qb][kb][r] = -INFINITY;
      float mx = fmaxf(fmaxf(sacc[qb][0][0], sacc[qb][0][1]), fmaxf(sacc[qb][0][2], sacc[qb][0][3]));
#pragma unroll
      for (int kb = 1; kb < KB; ++kb)
        mx = fmaxf(mx, fmaxf(fmaxf(sacc[qb][kb][0], sacc[qb][kb][1]), fmaxf(sacc[qb][kb][2], sacc[qb][kb][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mb = mx * sl2;
      float rs = 0.f;
      const half2_t ones2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
      for (int kb = 0; kb < KB + 1; ++kb)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          half2_t e = {(half_t)0.0f, (half_t)0.0f};
          if (kb < KB) {
            const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r], sl2, -mb));
            const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r + 1], sl2, -mb));
            e = half2_t{(half_t)e0, (half_t)e1};
            if (!ONES) rs = __builtin_amdgcn_fdot2(e, ones2, rs, false);  // sums exactly what the PV MFMA sees
          }
          pf[qb][kb >> 1][(kb & 1) * 4 + r] = e[0];
          pf[qb][kb >> 1][(kb & 1) * 4 + r + 1] = e[1];
        }
      l_sum[qb] = rs;
    }
    // ---- O^T = V^T . P^T
    f32x4 oacc[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int db = 0; db < 4; ++db) oacc[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int chunk = db * 2 + (p4 >> 1);
        // keys 208..223 of the last k-step do not exist: their probabilities are zero, so any finite V row serves
        const int r0 = ks * 32 + fg * 4 + q4, r1 = ks == 6 ? r0 : r0 + 16;
        const half4_t lo = lds_tr_read(&Vs[r0 * RW + ((chunk ^ (r0 & (CPR - 1))) << 3) + (p4 & 1) * 4]);
        const half4_t hi = lds_tr_read(&Vs[r1 * RW + ((chunk ^ (r1 & (CPR - 1))) << 3) + (p4 & 1) * 4]);
        const half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qb][ks], oacc[qb][db], 0, 0, 0);
      }
    }
    // ---- epilogue: lane owns O[q = fr][d = 16*db + 4*fg + r]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float l;
      if (ONES) {
        l = __shfl(oacc[qb][3][3], 48 + fr, 64);  // O[q][63] lives in lane group 3, register 3 of the last d-block
      } else {
        l = l_sum[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
      }
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
}


// ---- the same whole-sequence problem (128 < Tk <= 208, Tq <= 208, head dim <= 64) as a PERSISTENT kernel (round 3).
// attn_sp_kernel spends ~20 us per (window, head) item and workgroup (two co-resident) against ~1 us of MFMA time and ~3 us of
// HBM time for its 66 KB: every item pays a register-staged K/V load (two dependent rounds of global loads), a barrier, then
// per 32-query pair a global load of Q in front of the first MFMA — latencies that two co-resident workgroups only half hide.
// Here ONE 8-wave workgroup per CU walks a contiguous range of items:
//   * Q, K and V of item i+1 arrive by LDS-DMA (global_load_lds_dwordx4: 64-bit per-lane addresses, so window geometry and
//     the padding vectors need no second descriptor) into the other half of a double buffer (2 x 78 KB) while item i computes;
//     EXEC-masked lanes leave their LDS slots alone (tools/gll_probe.hip), which keeps the constant parts of the image — V's
//     ones column for head dim <= 56, zeros beyond the head dim and beyond Tk — written once at kernel start;
//   * the 13 query blocks are in flight at once: 7 waves take one pair each (the eighth only issues DMAs), Q fragments come
//     from LDS, so an item is ONE round of the pair code instead of two;
//   * one barrier per item; a wave confirms its own DMA pieces (s_waitcnt vmcnt(0)) just before it stores O, when they have
//     long landed, so no store latency sits in front of the barrier.
template <bool ONES>
__global__ __launch_bounds__(512, 1) void attn_spp_kernel(const lmx_attn_desc p, const Geo geo, const int items) {
  constexpr int RW = 64, CPR = 8, KB = 13, ROWS = KB * 16, QB = 2, NWV = 8;
  constexpr int TEN = ROWS * RW * 2;          // bytes of one tensor image: 26 KB
  constexpr int BUF = 3 * TEN;                // Q | K | V
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  // (kernel-argument fields the DMA plan selects between are copied to locals: a `cond ? p.a : p.b` on the argument struct makes
  // hipcc spill the whole struct to scratch and index it there)
  const int hd = p.hd, H_ = p.H, Tq_ = p.Tq, Tk_ = p.Tk, mode_ = geo.mode;
  const int64_t ldq_ = p.ldq, ldk_ = p.ldk, ldv_ = p.ldv;
  const int ws_ = geo.ws, wsq_ = geo.wsq, Gh_ = geo.Gh, Gw_ = geo.Gw, Gqh_ = geo.Gqh, Gqw_ = geo.Gqw, nWx_ = geo.nWx, nW_ = geo.nWy * geo.nWx;

  // ---- constant parts of both images: zeros everywhere, then V[:, 63] = 1 (the PV MFMA then accumulates the softmax sum)
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * BUF / 16; i += 512) *reinterpret_cast<u32x4*>(smem + i * 16) = z;
    __syncthreads();
    if (ONES)
      for (int i = tid; i < 2 * ROWS; i += 512) {
        const int bsel = i / ROWS, row = i - bsel * ROWS;
        half_t* v = reinterpret_cast<half_t*>(smem + bsel * BUF + 2 * TEN);
        v[row * RW + ((7 ^ (row & 7)) << 3) + 7] = (half_t)1.0f;
      }
  }

  // ---- this workgroup's items: XCD x owns a contiguous range (the heads of a window, and neighbouring windows, share its L2)
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int nwg = ((int)gridDim.x + 7 - xcd) >> 3;
  const int iq = items >> 3, ir = items & 7;
  const int ibase = xcd < ir ? xcd * (iq + 1) : ir * (iq + 1) + (xcd - ir) * iq;
  const int icnt = iq + (xcd < ir ? 1 : 0);

  // ---- DMA plan: tensor t (0 Q, 1 K, 2 V; compile-time in the issue code) is 26 wave-instructions of 8 rows; wave w issues
  // j = w, w + 8, w + 16, w + 24 (< 26) of every tensor; lane L -> row 8 j + (L >> 3), physical chunk L & 7 = logical chunk ^ (row & 7).
  // Everything that does not depend on the item is computed here, once: the byte offset of the lane's piece from the item's first
  // token (window origin resp. b * T) and the row's window coordinates for the padding test.  Per item a piece then costs one
  // 64-bit add (plus the padding test in the edge windows only).
  constexpr int NJ = 4;
  unsigned rel[3][NJ];  // byte offset from the item's origin in its tensor
  int tyx[3][NJ];       // (valid << 30) | (d << 20) | (ty << 10) | tx
#pragma unroll
  for (int ten = 0; ten < 3; ++ten)
#pragma unroll
    for (int m = 0; m < NJ; ++m) {
      const int j = wave + NWV * m;
      const int row = 8 * j + (lane >> 3);
      const int lc = (lane & 7) ^ (row & 7);
      const int d = lc * 8;
      const int T = ten == 0 ? Tq_ : Tk_;
      const int wsz = ten == 0 ? wsq_ : ws_;
      const int gw = ten == 0 ? Gqw_ : Gw_;
      const int64_t ld = ten == 0 ? ldq_ : (ten == 1 ? ldk_ : ldv_);
      const int ty = mode_ == 0 ? 0 : row / wsz, tx = mode_ == 0 ? row : row - ty * wsz;
      rel[ten][m] = (unsigned)((((int64_t)(mode_ == 0 ? row : ty * gw + tx)) * ld + d) * 2);
      tyx[ten][m] = ((j < 26 && row < T && d < hd) ? (1 << 30) : 0) | (d << 20) | (ty << 10) | tx;
    }
  const char* Qg = reinterpret_cast<const char*>(p.Q);
  const char* Kg = reinterpret_cast<const char*>(p.K);
  const char* Vg = reinterpret_cast<const char*>(p.V);
  const char* padk = reinterpret_cast<const char*>(p.pad_k);
  const char* padv = reinterpret_cast<const char*>(p.pad_v);
  // this lane's two query rows (fixed over the items): window coordinates for query_row without its divisions
  int qty[QB], qtx[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = (wave * QB + qb) * 16 + fr;
    qty[qb] = mode_ == 0 ? 0 : tq / wsq_;
    qtx[qb] = mode_ == 0 ? tq : tq - qty[qb] * wsq_;
  }

  auto issue = [&](int item, int bsel) {
    const int h = item % H_, b = item / H_;
    // item-uniform: origin of the item in each tensor, and how many rows / columns of its window lie on the grid
    int img = b, wy = 0, wx = 0;
    if (mode_ != 0) {
      img = b / nW_;
      const int w = b - img * nW_;
      wy = w / nWx_;
      wx = w - wy * nWx_;
    }
    const int64_t hcol = (int64_t)h * hd * 2;
    const char *oq, *ok, *ov;
    int qy = 1 << 20, qx = 1 << 20, ky = 1 << 20, kx = 1 << 20;  // rows / columns available (flat: unbounded)
    if (mode_ == 0) {
      oq = Qg + (int64_t)b * Tq_ * ldq_ * 2 + hcol;
      ok = Kg + (int64_t)b * Tk_ * ldk_ * 2 + hcol;
      ov = Vg + (int64_t)b * Tk_ * ldv_ * 2 + hcol;
    } else {
      const int64_t q0 = ((int64_t)img * Gqh_ + wy * wsq_) * Gqw_ + wx * wsq_;
      const int64_t k0 = ((int64_t)img * Gh_ + wy * ws_) * Gw_ + wx * ws_;
      oq = Qg + q0 * ldq_ * 2 + hcol;
      ok = Kg + k0 * ldk_ * 2 + hcol;
      ov = Vg + k0 * ldv_ * 2 + hcol;
      qy = Gqh_ - wy * wsq_, qx = Gqw_ - wx * wsq_, ky = Gh_ - wy * ws_, kx = Gw_ - wx * ws_;
    }
    const bool edge = qy < wsq_ || qx < wsq_ || ky < ws_ || kx < ws_;  // uniform: interior windows skip every padding test
    char* base = smem + bsel * BUF + wave * 1024;
#pragma unroll
    for (int ten = 0; ten < 3; ++ten) {
      const char* org = ten == 0 ? oq : (ten == 1 ? ok : ov);
      const char* pv = (ten == 1 ? padk : padv) + hcol;
#pragma unroll
      for (int m = 0; m < NJ; ++m) {
        if (wave + NWV * m >= 26) break;
        const int pl = tyx[ten][m];
        bool act = (pl >> 30) & 1;
        const char* src = org + rel[ten][m];
        if (edge) {
          const int ty = (pl >> 10) & 1023, tx = pl & 1023;
          if (ten == 0) {
            act = act && ty < qy && tx < qx;  // a padded query: nothing to load, nothing is stored for it
          } else if (ty >= ky || tx >= kx) {
            src = pv + ((pl >> 20) & 1023) * 2;  // a padded key: the qkv bias (what Linear(0) yields)
          }
        }
        char* dst = base + ten * TEN + m * (NWV * 1024);  // wave-uniform; lane L lands at + 16 L
        if (act) __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src), (lds_ptr_t)dst, 16, 0, 0);
      }
    }
  };

  const float sl2 = p.scale * 1.44269504088896340736f;
  const int q4 = fr >> 2, p4 = fr & 3;
  const int nqb = (Tq_ + 15) / 16, npair = (nqb + QB - 1) / QB;
  __syncthreads();  // the constant parts are in place before the first DMA lands on top of them
  // Issuing a wave's ~10 LDS-DMA pieces blocks THAT wave for ~3k cycles (a wave issues one global_load_lds per ~300 cycles: a
  // lone producer wave needs 22k cycles for the 78 pieces of an item, profiles/r03_attn_spp.txt) but not its SIMD: the two waves
  // of a SIMD therefore issue at DIFFERENT points of the item — waves 0 - 3 before their S phase, waves 4 - 7 after it — so that
  // one of them computes while the other sits in the vector-memory queue.
  const bool early = wave < NWV / 2;
  // (the second wave of each SIMD (4 - 6) is the YOUNGER one and loses the instruction arbitration by age: softmax 3.1k cycles
  // against 1.2k for its partner.  s_setprio 2 on it swaps the roles and leaves the item period at 16k cycles: the SIMD's issue
  // slots are what is exhausted — ~3300 instructions per item and SIMD, of which 216 are MFMAs — not one wave's share of them.)
  if (jx < icnt) issue(ibase + jx, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int bsel = 0;
#ifdef LMX_DBG_TIMELINE
  int tl_it = -1;
#endif
  for (int li = jx; li < icnt; li += nwg, bsel ^= 1) {
    const int item = ibase + li;
    const int h = item % p.H, b = item / p.H;
#ifdef LMX_DBG_TIMELINE
    ++tl_it;
#endif
    ATL(0);
    __syncthreads();  // every wave's pieces of this item have landed; every wave is done with the previous item's image
    ATL(1);
    const bool more = li + nwg < icnt;
    if (more && (early || wave >= npair)) issue(item + nwg, bsel ^ 1);
    ATL(2);
    const half_t* Qs = reinterpret_cast<const half_t*>(smem + bsel * BUF);
    const half_t* Ks = Qs + ROWS * RW;
    const half_t* Vs = Ks + ROWS * RW;
    const int pr = wave;
    if (pr < npair) {
      half8_t qf[QB][2];
      int64_t qrow[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const int tq = (pr * QB + qb) * 16 + fr;
        if (mode_ == 0) {
          qrow[qb] = tq < Tq_ ? (int64_t)b * Tq_ + tq : -1;
        } else {
          const int img = b / nW_, w = b - img * nW_;
          const int wy = w / nWx_, wx = w - wy * nWx_;
          const int y = wy * wsq_ + qty[qb], x = wx * wsq_ + qtx[qb];
          qrow[qb] = (tq < Tq_ && y < Gqh_ && x < Gqw_) ? ((int64_t)img * Gqh_ + y) * Gqw_ + x : -1;
        }
        const int lr = tq < ROWS ? tq : ROWS - 1;  // (blocks past the image read a valid row; their results are never stored)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = *reinterpret_cast<const half8_t*>(&Qs[lr * RW + ((((ks << 2) + fg) ^ (lr & 7)) << 3)]);
      }
      f32x4 sacc[QB][KB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) sacc[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
      // K fragments through a three-deep register ring, two reads ahead of the MFMAs that consume them: left to itself hipcc
      // re-uses ONE fragment register and emits read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs 26 times, i.e. an LDS latency per 32
      // cycles of matrix work (the phase measured 3.0k cycles for 0.83k of MFMA: profiles/r03_attn_spp.txt)
      {
        auto kread = [&](int i) {  // i = ks * KB + kb
          const int ks = i / KB, kb = i - ks * KB;
          return *reinterpret_cast<const half8_t*>(&Ks[(kb * 16 + fr) * RW + ((((ks << 2) + fg) ^ (fr & (CPR - 1))) << 3)]);
        };
        half8_t kring[3];
        kring[0] = kread(0);
        kring[1] = kread(1);
#pragma unroll
        for (int i = 0; i < 2 * KB; ++i) {
          if (i + 2 < 2 * KB) kring[(i + 2) % 3] = kread(i + 2);
          __builtin_amdgcn_sched_barrier(0);  // keep the read ahead: nothing moves across
          const int ks = i / KB, kb = i - ks * KB;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kring[i % 3], qf[qb][ks], sacc[qb][kb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (more && !early) issue(item + nwg, bsel ^ 1);
      ATL(3);
      half8_t pf[QB][7];
      float l_sum[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int kb = 8; kb < KB; ++kb)
          if (kb * 16 + 16 > Tk_) {  // (uniform: only the block that straddles Tk, and those past it, pay for the test)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kb * 16 + fg * 4 + r >= Tk_) sacc[qb][kb][r] = -INFINITY;
          }
        // row maximum: two v_max3_f32 chains (fmaxf on MFMA results costs a canonicalising v_max per operand), then the VALU
        // cross-lane step of attn_gp_kernel instead of two ds_bpermute round trips
        float ma = vmax3(sacc[qb][0][0], sacc[qb][0][1], sacc[qb][0][2]);
        float mc = vmax3(sacc[qb][0][3], sacc[qb][1][0], sacc[qb][1][1]);
        ma = vmax3(ma, sacc[qb][1][2], sacc[qb][1][3]);
#pragma unroll
        for (int kb = 2; kb < KB; ++kb) {
          if (kb & 1)
            ma = vmax3(ma, sacc[qb][kb][0], sacc[qb][kb][1]), ma = vmax3(ma, sacc[qb][kb][2], sacc[qb][kb][3]);
          else
            mc = vmax3(mc, sacc[qb][kb][0], sacc[qb][kb][1]), mc = vmax3(mc, sacc[qb][kb][2], sacc[qb][kb][3]);
        }
        const float mx = row_max4(ma, mc);
        const float mb = mx * sl2;
        float rs = 0.f;
        const half2_t ones2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
        for (int kb = 0; kb < KB + 1; ++kb)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            half2_t e = {(half_t)0.0f, (half_t)0.0f};
            if (kb < KB) {
              const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r], sl2, -mb));
              const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r + 1], sl2, -mb));
              e = half2_t{(half_t)e0, (half_t)e1};
              if (!ONES) rs = __builtin_amdgcn_fdot2(e, ones2, rs, false);
            }
            pf[qb][kb >> 1][(kb & 1) * 4 + r] = e[0];
            pf[qb][kb >> 1][(kb & 1) * 4 + r + 1] = e[1];
          }
        l_sum[qb] = rs;
      }
      ATL(4);
      f32x4 oacc[QB][4];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int db = 0; db < 4; ++db) oacc[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
      {  // V^T fragments (two transposing 8-byte reads each) through the same kind of ring.  The reads are inline asm with counted
        // waits of their own: through the builtin hipcc cannot tell them from the LDS-DMA writes in flight (the NEXT item's pieces
        // this wave has just issued) and waits for vmcnt(0) in front of the first one — the prefetch then has to land before PV starts.
        u32x2_t vlo[3], vhi[3];
        auto vread = [&](int i, u32x2_t& lo, u32x2_t& hi) {  // i = ks * 4 + db
          const int ks = i >> 2, db = i & 3;
          const int chunk = db * 2 + (p4 >> 1);
          const int r0 = ks * 32 + fg * 4 + q4, r1 = ks == 6 ? r0 : r0 + 16;
          const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const half_t*)&Vs[r0 * RW + ((chunk ^ (r0 & (CPR - 1))) << 3) + (p4 & 1) * 4];
          const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const half_t*)&Vs[r1 * RW + ((chunk ^ (r1 & (CPR - 1))) << 3) + (p4 & 1) * 4];
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1) : "memory");
        };
        vread(0, vlo[0], vhi[0]);
        vread(1, vlo[1], vhi[1]);
#pragma unroll
        for (int i = 0; i < 28; ++i) {
          if (i + 2 < 28) vread(i + 2, vlo[(i + 2) % 3], vhi[(i + 2) % 3]);
          // fragment i has arrived once only the younger reads are outstanding (the registers are operands of the wait, so that no
          // consumer can be scheduled in front of it)
          if (i + 2 < 28)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vlo[i % 3]), "+v"(vhi[i % 3])::"memory");
          else if (i + 1 < 28)
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vlo[i % 3]), "+v"(vhi[i % 3])::"memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[i % 3]), "+v"(vhi[i % 3])::"memory");
          const u32x4_t w = {vlo[i % 3][0], vlo[i % 3][1], vhi[i % 3][0], vhi[i % 3][1]};
          const half8_t vf = __builtin_bit_cast(half8_t, w);
          const int ks = i >> 2, db = i & 3;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qb][ks], oacc[qb][db], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // my DMA pieces of the NEXT item before my stores: they have long landed, and no store sits in front of the next wait
      ATL(5);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ATL(6);
      half_t* O = reinterpret_cast<half_t*>(p.O);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float l;
        if (ONES) {
          l = __shfl(oacc[qb][3][3], 48 + fr, 64);
        } else {
          l = l_sum[qb];
          l += __shfl_xor(l, 16, 64);
          l += __shfl_xor(l, 32, 64);
        }
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
      ATL(7);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ATL(7);
    }
  }
}

}  // namespace

static int build_geo(const lmx_attn_desc& d, Geo& g);

extern "C" int lmx_k_relpos_tables(const lmx_attn_desc* dp, const float* rel_pos_h, const float* rel_pos_w, int S, void* out,
                                   lmx_stream_t stream) {
  LMX_REQUIRE(dp && rel_pos_h && rel_pos_w && out, "lmx_k_relpos_tables: null pointer");
  const lmx_attn_desc& d = *dp;
  LMX_REQUIRE(d.Q && d.B > 0 && d.H > 0 && d.Tq > 0 && d.hd % 8 == 0 && d.hd <= 96 && d.ldq % 8 == 0 && aligned16(d.Q),
              "lmx_k_relpos_tables: descriptor");
  LMX_REQUIRE(S > 0 && S * S == d.Tq, "lmx_k_relpos_tables: S=%d vs Tq=%d", S, d.Tq);
  Geo g{};
  const int rc = build_geo(d, g);
  if (rc) return rc;
  // padded queries of a window (row < 0) produce zeros: q is read as zero
  const int64_t items = (int64_t)d.B * d.H * 2 * S;
  const int64_t grid = (items + 3) / 4;
  LMX_REQUIRE(grid < (1ll << 31) && aligned16(rel_pos_h) && aligned16(rel_pos_w), "lmx_k_relpos_tables: size / alignment");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d.hd <= 64)
    hipLaunchKernelGGL((relpos_tables_kernel<2>), dim3((unsigned)grid), dim3(256), 0, st, d, g, rel_pos_h, rel_pos_w, S,
                       reinterpret_cast<half_t*>(out), items);
  else
    hipLaunchKernelGGL((relpos_tables_kernel<3>), dim3((unsigned)grid), dim3(256), 0, st, d, g, rel_pos_h, rel_pos_w, S,
                       reinterpret_cast<half_t*>(out), items);
  return lmx_launch_check("relpos_tables_kernel");
}

extern "C" int lmx_k_attention(const lmx_attn_desc* dp, lmx_stream_t stream) {
  LMX_REQUIRE(dp != nullptr, "lmx_k_attention: null descriptor");
  const lmx_attn_desc& d = *dp;
  LMX_REQUIRE(d.Q && d.K && d.V && d.O, "lmx_k_attention: null Q/K/V/O");
  LMX_REQUIRE(d.B > 0 && d.H > 0 && d.Tq > 0 && d.Tk > 0, "lmx_k_attention: empty problem");
  LMX_REQUIRE(d.hd % 8 == 0 && d.hd > 0 && d.hd <= 96, "lmx_k_attention: head dim %d (need multiple of 8, <=96)", d.hd);
  LMX_REQUIRE(d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 4 == 0, "lmx_k_attention: strides");
  LMX_REQUIRE(aligned16(d.Q) && aligned16(d.K) && aligned16(d.V) && ((((uintptr_t)d.O) & 7) == 0),
              "lmx_k_attention: alignment");
  Geo g{};
  {
    const int rc = build_geo(d, g);
    if (rc) return rc;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  if (d.Tq <= 16 && d.Tk <= 16 && !d.rel && d.hd <= 64) {  // one wave per (batch | window, head)
    const int64_t items = (int64_t)d.B * d.H;
    LMX_REQUIRE(items < (1ll << 31), "lmx_k_attention: grid too large");
    hipLaunchKernelGGL(attn_small_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, d, g, (int)items);
    return lmx_launch_check("attn_small_kernel");
  }
  static int no_sp = -1;
  if (no_sp < 0) no_sp = getenv("LMX_ATTN_NO_SP") ? 1 : 0;
  if (!no_sp && !d.rel && d.hd <= 64 && d.Tk > 128 && d.Tk <= 208 && d.Tq > 64 && d.Tq <= 208) {  // whole sequence per workgroup
    const int64_t items = (int64_t)d.B * d.H;
    LMX_REQUIRE(items < (1ll << 31), "lmx_k_attention: grid too large");
    static int sp_qb = 0, no_spp = -1;
    if (!sp_qb) sp_qb = getenv("LMX_ATTN_SP_QB") ? atoi(getenv("LMX_ATTN_SP_QB")) : 2;
    if (no_spp < 0) no_spp = getenv("LMX_ATTN_NO_SPP") ? 1 : 0;
    // persistent double-buffered form (one 8-wave workgroup per CU walks a range of items): window geometry needs both padding
    // vectors (a padded key with no vector would have to be WRITTEN as zeros, which the masked DMA does not do)
    if (!no_spp && items >= 64 && (d.mode == 0 || (d.pad_k && d.pad_v)) && d.hd % 8 == 0) {
      constexpr int SMEM = 2 * 3 * 208 * 64 * 2;
      static bool attr_set = false;
      if (!attr_set) {
        LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_spp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_spp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
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
      const unsigned grid = (unsigned)(items < n_cu ? items : n_cu);
      if (d.hd <= 56)
        hipLaunchKernelGGL((attn_spp_kernel<true>), dim3(grid), dim3(512), SMEM, st, d, g, (int)items);
      else
        hipLaunchKernelGGL((attn_spp_kernel<false>), dim3(grid), dim3(512), SMEM, st, d, g, (int)items);
      return lmx_launch_check("attn_spp_kernel");
    }
    if (d.hd <= 56 && sp_qb == 2)
      hipLaunchKernelGGL((attn_sp_kernel<2, true>), dim3((unsigned)items), dim3(256), 0, st, d, g);
    else if (d.hd <= 56)
      hipLaunchKernelGGL((attn_sp_kernel<1, true>), dim3((unsigned)items), dim3(256), 0, st, d, g);
    else if (sp_qb == 2)
      hipLaunchKernelGGL((attn_sp_kernel<2, false>), dim3((unsigned)items), dim3(256), 0, st, d, g);
    else
      hipLaunchKernelGGL((attn_sp_kernel<1, false>), dim3((unsigned)items), dim3(256), 0, st, d, g);
    return lmx_launch_check("attn_sp_kernel");
  }
  const bool wide = d.hd > 64;  // SAM ViT-H (head dim 80): 128-half LDS rows, one 16-query block per wave
  const bool big = d.Tq > 64 && !wide;
  const int qtile = big ? 128 : 64;
  int nQT = (d.Tq + qtile - 1) / qtile;
  const int64_t nblk = (int64_t)d.B * d.H * nQT;
  LMX_REQUIRE(nblk < (1ll << 31), "lmx_k_attention: grid too large");
  const bool ones = d.hd <= 56;
  // LDS-DMA staging: flat geometry without bias, at least a few key tiles, per-(batch) K/V extent within a 31-bit buffer
  static int no_dma = -1, no_lazy = 0, no_gp = 0;
  if (no_dma < 0) {
    no_gp = getenv("LMX_ATTN_NO_GP") ? 1 : 0;  // A/B: the unpipelined LDS-DMA form of attn_kernel
    no_dma = getenv("LMX_ATTN_NO_DMA") ? 1 : 0;
    no_lazy = getenv("LMX_ATTN_NO_LAZY") ? 1 : 0;
  }
  const bool dma = !no_dma && !wide && d.mode == 0 && !d.rel && big && d.Tk >= 256 && (int64_t)d.Tk * d.ldk * 2 < 0x7fff0000ll &&
                   (int64_t)d.Tk * d.ldv * 2 < 0x7fff0000ll;
  if (no_lazy) nQT = -nQT;
  // the pipelined kernel runs 4 waves (128 queries) per workgroup; LMX_ATTN_GP_WAVES=8 selects the 8-wave form (256 queries share a
  // K / V tile, one LDS-DMA piece per wave and tensor instead of two): measured SLOWER, 1595 vs 1526 us at B = 30, H = 8, T = 4096 —
  // the barrier among eight waves costs more than the saved issue slots (identical bits)
  static int gp_waves = -1;
  if (gp_waves < 0) gp_waves = getenv("LMX_ATTN_GP_WAVES") ? atoi(getenv("LMX_ATTN_GP_WAVES")) : 4;
  const bool gp8 = gp_waves == 8;
  const int nQT8 = (d.Tq + 255) / 256;
#define GP_LAUNCH(ones)                                                                                                                    \
  do {                                                                                                                                     \
    if (gp8)                                                                                                                               \
      hipLaunchKernelGGL((attn_gp_kernel<ones, 8>), dim3((unsigned)((int64_t)d.B * d.H * nQT8)), dim3(512), 2 * 3 * 64 * 64 * 2, st, d,    \
                         no_lazy ? -nQT8 : nQT8);                                                                                          \
    else                                                                                                                                   \
      hipLaunchKernelGGL((attn_gp_kernel<ones, 4>), dim3((unsigned)nblk), dim3(256), 2 * 3 * 64 * 64 * 2, st, d, nQT);                     \
  } while (0)
  if (d.rel) {
    LMX_REQUIRE(d.rel_S > 0 && d.rel_S * d.rel_S == d.Tk && d.Tq == d.Tk, "lmx_k_attention: rel_S=%d does not match Tq=%d Tk=%d",
                d.rel_S, d.Tq, d.Tk);
    if (wide)
      hipLaunchKernelGGL((attn_kernel<1, false, true, false, 96>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
    else if (big)
      hipLaunchKernelGGL((attn_kernel<2, false, true, false, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
    else
      hipLaunchKernelGGL((attn_kernel<1, false, true, false, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  } else if (wide)
    hipLaunchKernelGGL((attn_kernel<1, false, false, false, 96>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  else if (dma && !no_gp && ones)
    GP_LAUNCH(true);
  else if (dma && !no_gp)
    GP_LAUNCH(false);
  else if (dma && ones)
    hipLaunchKernelGGL((attn_kernel<2, true, false, true, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  else if (dma)
    hipLaunchKernelGGL((attn_kernel<2, false, false, true, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  else if (big && ones)
    hipLaunchKernelGGL((attn_kernel<2, true, false, false, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  else if (big)
    hipLaunchKernelGGL((attn_kernel<2, false, false, false, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  else if (ones)
    hipLaunchKernelGGL((attn_kernel<1, true, false, false, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  else
    hipLaunchKernelGGL((attn_kernel<1, false, false, false, 64>), dim3((unsigned)nblk), dim3(256), 0, st, d, g, nQT);
  return lmx_launch_check("attn_kernel");
}

static int build_geo(const lmx_attn_desc& d, Geo& g) {
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
  return LMX_OK;
}
