// mlp.hip — LayerNorm, then fused fc1 -> GELU -> fc2 -> +residual for NARROW token widths (Hiera stages 1-2: D = 112, 224;
// TF:models/sam2/modeling_sam2.py Sam2MultiScaleBlock.forward: `hidden_states + mlp(layer_norm2(hidden_states))`).
//
// Why: at D <= 224 the unfused chain  LN kernel -> GEMM(D->4D, GELU) -> GEMM(4D->D, +res)  is HBM-bound on its
// intermediates: per token it moves 4D (LN read) + 2D (LN write) + 2D + 8D (fc1) + 8D + 4D + 4D (fc2) = 32 D bytes, of
// which only the f32 residual stream (4D read + 4D write) is algorithmic.  Here a workgroup keeps its tokens' 4D hidden
// activations in registers: the structure is the flash-attention one (attn.hip) with W1 rows in the role of keys, W2
// columns in the role of values and GELU in the role of the softmax:
//     H^T[32 hidden][tokens] = W1c . LN(x)^T      (MFMA, A = W1 rows from LDS, B = the wave's normalised tokens, registers)
//     P = f16(gelu(H^T + b1))                      (the accumulator layout IS the next B operand, k-slot permuted)
//     O^T[D][tokens]        += W2c^T-slice . P     (MFMA, A = W2 rows from LDS read as two 8-byte pieces)
// The weights (<= 800 KB, L2-resident) stream through a 4-slot LDS ring in 32-hidden-unit chunks by LDS-DMA
// (buffer_load ... lds, no VGPR round trip), three chunks in flight, ONE raw s_barrier + counted s_waitcnt vmcnt per
// chunk exactly as gemm2.hip.  (A first version staged through registers one chunk ahead: every chunk then waited a full
// L2 latency for the loads issued at its own start — 6000 cycles per chunk against 480 cycles of MFMA.)
// HBM traffic per token: LayerNorm 4D read + 2D write, fused kernel 2D + 4D read + 4D write (the residual enters as the
// initial fc2 accumulator) = 16 D bytes against 32 D for the unfused chain.
#include "common.h"
#include <stdlib.h>

#ifndef LMX_MLP_DBG
#define LMX_MLP_DBG 0
#endif
namespace lmx_mlp {

constexpr int HC = 32;   // hidden units per chunk (one MFMA k-step of the second GEMM)

// D: token width (multiple of 16, <= 512).  NW waves of QB 16-token blocks: a workgroup owns NW*16*QB tokens.
// Every wave issues PT LDS-DMA instructions (1 KB each) per chunk: NW*PT KB = one ring slot = [W1 chunk | W2 chunk].
// NST: LDS ring slots (NST-1 chunks in flight).  OCC: workgroups per CU the register budget must allow.
// INLN: the LayerNorm runs in this kernel (row statistics reduced over the four lanes of a token) instead of reading the f16
// rows a LayerNorm launch wrote: saves that launch and 4D bytes per token.  Round 1 had to take it out — a few hundred rows
// per million got run-to-run different statistics whenever MFMA waves shared the SIMD; round 2 traced that to the SLP
// vectoriser's v_pk_add_f32 op_sel:[0,1] in the horizontal sums, which the build now forbids.
template <int D, int QB, int NW, int NST, int OCC, bool INLN, bool RING = false>
__global__ __launch_bounds__(NW * 64, OCC * NW / 4) void ln_mlp_kernel(float* __restrict__ x, int64_t ldx,
                                                                          const half_t* __restrict__ hn,
                                                                          const float* __restrict__ gam,
                                                                          const float* __restrict__ bet, const float eps,
                                                                          const half_t* __restrict__ w1,
                                                                          const float* __restrict__ b1,
                                                                          const half_t* __restrict__ w2,
                                                                          const float* __restrict__ b2, int64_t rows,
                                                                          half_t* __restrict__ x16,
                                                                          const float* __restrict__ gam_n,
                                                                          const float* __restrict__ bet_n,
                                                                          half_t* __restrict__ h_n, const int stagger) {
  constexpr int KS = (D + 31) / 32;           // k-steps of the first GEMM
  constexpr int DP = D <= 128 ? 128 : (D <= 256 ? 256 : 512);  // halfs per LDS row of a W1 chunk (power of two: XOR swizzle stays in the row)
  constexpr int DB = D / 16;                  // 16-wide output blocks
  constexpr int NCH = 4 * D / HC;             // chunks
  constexpr int W1_BYTES = HC * DP * 2;       // 8 KB | 16 KB
  constexpr int W2_ROWS = DP;                 // W2 chunk rows padded to 128 | 256 (rows >= D read as zeros, never used)
  constexpr int W2_BYTES = W2_ROWS * 64;      // 8 KB | 16 KB
  constexpr int STAGE = W1_BYTES + W2_BYTES;
  constexpr int PT = STAGE / (NW * 1024);     // DMA instructions per wave per chunk (4, or 2 for D = 112 on 8 waves)
  constexpr int LA = NST - 1;
  static_assert(STAGE == NW * PT * 1024 && PT >= 1, "one ring slot = NW waves x PT KB");
  static_assert(W1_BYTES == (NW / 2) * PT * 1024, "the first half of the waves stages W1, the second half W2");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* b1s = reinterpret_cast<float*>(smem + NST * STAGE);  // fc1 bias, 4D floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- LDS-DMA plan.  An instruction writes 64 x 16 B linearly, so the swizzle goes on the per-lane SOURCE address;
  // lanes whose piece does not exist (k-padding of W1, row padding of W2) take an out-of-range offset: the buffer
  // descriptor's range check returns zeros.
  const __amdgpu_buffer_rsrc_t w1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(w1), 0, 4 * D * D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(w2), 0, 4 * D * D * 2, 0x00020000);
  const bool stage_w1 = wave < NW / 2;
  const unsigned OOB = 0x80000000u;
  unsigned voff[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    if (stage_w1) {
      constexpr int CPR = DP / 8;                       // 16-byte pieces per LDS row: 16 | 32
      const int i = wave * PT + t;                      // KB index inside the W1 chunk
      const int row = i * (64 / CPR) + lane / CPR;      // 4 | 2 rows per instruction
      const int lc = (lane % CPR) ^ (row & 15);         // logical piece stored at this physical slot
      voff[t] = lc < D / 8 ? (unsigned)(row * D * 2 + lc * 16) : OOB;
    } else {
      const int i = (wave - NW / 2) * PT + t;           // KB index inside the W2 chunk: 16 rows of 64 B
      const int row = i * 16 + (lane >> 2);
      const int lc = (lane & 3) ^ ((row >> 2) & 3);
      voff[t] = row < D ? (unsigned)(row * (4 * D) * 2 + lc * 16) : OOB;
    }
  }
  auto issue = [&](int c, int slot) {
    char* dst = smem + slot * STAGE + wave * (PT * 1024);  // waves NW/2.. land in the W2 half: W1_BYTES == NW/2 * 4 KB
    const int soff = stage_w1 ? c * (HC * D * 2) : c * (HC * 2);
#pragma unroll
    for (int t = 0; t < PT; ++t) lds_dma16(stage_w1 ? w1_rs : w2_rs, dst + t * 1024, voff[t], soff);
  };
#pragma unroll
  for (int c = 0; c < LA; ++c) issue(c, c);

  for (int i = tid; i < 4 * D; i += NW * 64) b1s[i] = b1[i];

  // ---- this wave's normalised tokens (f16, written by the LayerNorm kernel the entry point launches first) as MFMA B-operand
  // fragments: lane (fr, fg) holds, for token fr of block qb, the 8 features 32*ks + 8*fg .. +7 of every k-step.
  // (The first version normalised in here, reducing over the four lanes of a token with __shfl_xor: whenever waves of other
  // workgroups shared the SIMD, a few hundred rows per million got run-to-run different statistics — with ds_bpermute and
  // with v_permlane swaps alike, never with one workgroup per CU, cause not identified (tools/mlp_selfcheck.py).  The
  // standalone LayerNorm kernel is bit-reproducible in every configuration tested, so the reduction lives there.)
  const int64_t tok0 = (int64_t)blockIdx.x * (NW * 16 * QB) + wave * (16 * QB);
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  half8_t xn[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int64_t t = tok0 + qb * 16 + fr;
    if constexpr (INLN) {
      // LayerNorm straight into MFMA B-operand fragments: a token's D features sit in the 4 lanes fr, fr+16, fr+32, fr+48.
      // Three passes over the (L1/L2-hot) row instead of holding its D/4 values per lane keep the prologue's register peak
      // below the main loop's.
      const float* xr = x + (t < rows ? t : 0) * ldx;
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d = ks * 32 + fg * 8;
        if (d < D) {  // D % 8 == 0: a lane's 8 features are all in or all out
          const f32x4 a = *reinterpret_cast<const f32x4*>(xr + d), b = *reinterpret_cast<const f32x4*>(xr + d + 4);
          s += (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]);
        }
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d = ks * 32 + fg * 8;
        if (d < D) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(xr + d) - mean, b = *reinterpret_cast<const f32x4*>(xr + d + 4) - mean;
          q += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]) + (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
        }
      }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / D) + eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d = ks * 32 + fg * 8;
        const bool ok = d < D;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam + (ok ? d : 0)), g1 = *reinterpret_cast<const f32x4*>(gam + (ok ? d + 4 : 0));
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bet + (ok ? d : 0)), c1 = *reinterpret_cast<const f32x4*>(bet + (ok ? d + 4 : 0));
        const f32x4 va = *reinterpret_cast<const f32x4*>(xr + (ok ? d : 0)), vb = *reinterpret_cast<const f32x4*>(xr + (ok ? d + 4 : 0));
        half8_t hh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hh[e] = (half_t)(ok ? (va[e] - mean) * rstd * g0[e] + c0[e] : 0.f);
          hh[4 + e] = (half_t)(ok ? (vb[e] - mean) * rstd * g1[e] + c1[e] : 0.f);
        }
        xn[qb][ks] = hh;
      }
    } else {
      // the normalised tokens (f16) written by the LayerNorm launch that precedes this kernel
      const half_t* hr = hn + (t < rows ? t : 0) * D;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d = ks * 32 + fg * 8;
        const bool ok = d < D;  // D % 8 == 0: a lane's 8 features are all in or all out
        const half8_t v = *reinterpret_cast<const half8_t*>(hr + (ok ? d : 0));
        xn[qb][ks] = ok ? v : zero8;
      }
    }
  }

  // the fc2 accumulators start from x + b2 (the residual and the bias ride the MFMA chain): x is read from HBM ONCE —
  // these loads hit the lines the LayerNorm passes just pulled into L1/L2 — and the epilogue is a pure store
  __builtin_amdgcn_sched_barrier(0);  // keep these loads behind the LayerNorm: hoisted, they raise its register peak into spills
  f32x4 oacc[QB][DB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int64_t t = tok0 + qb * 16 + fr;
    const float* xr = x + (t < rows ? t : 0) * ldx;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int d = db * 16 + fg * 4;
      oacc[qb][db] = *reinterpret_cast<const f32x4*>(xr + d) + *reinterpret_cast<const f32x4*>(b2 + d);
    }
  }

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // b1s: a raw s_barrier does not wait for this wave's LDS writes
  for (int c = 0; c < NCH; ++c) {
    // chunk c has landed once only this wave's DMAs of the (at most LA-1) younger chunks are outstanding; everything the
    // prologue loaded from global memory is consumed by now, and nothing else is loaded from global memory inside the
    // loop (the fc1 bias sits in LDS), so the count is exact
    const int left = NCH - 1 - c;
    if (LA >= 3 && left >= 2)
      wait_vmcnt<(LA >= 3 ? 2 : 0) * PT>();
    else if (LA >= 2 && left >= 1)
      wait_vmcnt<(LA >= 2 ? 1 : 0) * PT>();
    else
      wait_vmcnt<0>();  // (a two-slot ring, LA = 1: only chunk c itself is outstanding here)
    __builtin_amdgcn_s_barrier();  // (also publishes b1s on the first pass)
    // Issuing a wave's PT LDS-DMA pieces blocks THAT wave for ~300 cycles per piece (profiles/r03_attn_spp.txt) but not its SIMD.
    // stagger: the two waves of a SIMD (w and w + NW/2) issue at different points of the chunk — the first before its fc1 MFMAs, the
    // second after them — so that one computes while the other sits in the vector-memory queue.  (The slot being refilled was
    // last read in the previous iteration, which the barrier above has closed: any point of this iteration is safe.)
    const bool late_issue = stagger && wave >= NW / 2;
    if (!late_issue && c + LA < NCH) issue(c + LA, (c + LA) % NST);
    const char* st = smem + (c % NST) * STAGE;
    const half_t* W1c = reinterpret_cast<const half_t*>(st);
    const char* W2c = st + W1_BYTES;

    // fc1 bias of this lane's 8 hidden units: 32c + 16hb + 4fg + i; the MFMA chain adds it for free
    f32x4 sacc[QB][2];
    {
      const f32x4 bia0 = *reinterpret_cast<const f32x4*>(b1s + c * HC + fg * 4);
      const f32x4 bia1 = *reinterpret_cast<const f32x4*>(b1s + c * HC + 16 + fg * 4);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        sacc[qb][0] = bia0;
        sacc[qb][1] = bia1;
      }
    }
    // ---- H^T = W1c . LN(x)^T
    if constexpr (RING) {
      // W1 fragments through a three-deep register ring, two reads ahead of their MFMAs: left alone hipcc (at the register cap)
      // emits read -> s_waitcnt -> MFMAs per fragment, an LDS latency per 16 - 32 cycles of matrix work (profiles/r03_fused_mlp_d448.txt)
      auto w1read = [&](int i) {  // i = ks * 2 + hb
        const int ks = i >> 1, row = (i & 1) * 16 + fr;
        return *reinterpret_cast<const half8_t*>(W1c + row * DP + (((ks * 4 + fg) ^ (row & 15)) << 3));
      };
      half8_t ring[3];
      ring[0] = w1read(0);
      ring[1] = w1read(1);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        if (i + 2 < 2 * KS) ring[(i + 2) % 3] = w1read(i + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          if (LMX_MLP_DBG != 2 || rows < 0) sacc[qb][i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[i % 3], xn[qb][i >> 1], sacc[qb][i & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int row = hb * 16 + fr;
        const half8_t a = *reinterpret_cast<const half8_t*>(W1c + row * DP + (((ks * 4 + fg) ^ (row & 15)) << 3));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][hb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xn[qb][ks], sacc[qb][hb], 0, 0, 0);
      }
    }
    }
    if (late_issue && c + LA < NCH) issue(c + LA, (c + LA) % NST);
    // ---- P = f16(gelu(H^T)): accumulator element i of block hb is hidden unit 16hb + 4fg + i of token fr — exactly
    // k-slot 8fg + (4hb + i) of the next MFMA's B operand once W2's columns are read in the same permuted order
    half8_t pf[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
#if LMX_MLP_DBG == 1  // decomposition build: no GELU (wrong results)
        const f32x2 g0 = f32x2{sacc[qb][hb][0], sacc[qb][hb][1]};
        const f32x2 g1 = f32x2{sacc[qb][hb][2], sacc[qb][hb][3]};
#else
        const f32x2 g0 = gelu_pk(f32x2{sacc[qb][hb][0], sacc[qb][hb][1]});
        const f32x2 g1 = gelu_pk(f32x2{sacc[qb][hb][2], sacc[qb][hb][3]});
#endif
        pf[qb][4 * hb + 0] = (half_t)g0[0];
        pf[qb][4 * hb + 1] = (half_t)g0[1];
        pf[qb][4 * hb + 2] = (half_t)g1[0];
        pf[qb][4 * hb + 3] = (half_t)g1[1];
      }
    }
    // ---- O^T += W2[:, chunk] . P.  W2 chunk row = 64 B = four 16-B pieces, piece p stored at p ^ ((row >> 2) & 3);
    // the fragment is hidden units 4fg..4fg+3 (piece fg>>1) and 16+4fg..+3 (piece 2 + (fg>>1)), 8 bytes each
    auto w2read = [&](int db) {
      const int row = db * 16 + fr;
      const int sw = (row >> 2) & 3;
      const char* wr = W2c + row * 64 + (fg & 1) * 8;
      const half4_t lo = *reinterpret_cast<const half4_t*>(wr + (((fg >> 1)) ^ sw) * 16);
      const half4_t hi = *reinterpret_cast<const half4_t*>(wr + ((2 + (fg >> 1)) ^ sw) * 16);
      return half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    if constexpr (RING) {
      half8_t ring[3];
      ring[0] = w2read(0);
      ring[1] = w2read(1);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if (db + 2 < DB) ring[(db + 2) % 3] = w2read(db + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          if (LMX_MLP_DBG != 3 || rows < 0) oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[db % 3], pf[qb], oacc[qb][db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const half8_t a = w2read(db);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[qb], oacc[qb][db], 0, 0, 0);
      }
    }
  }

  // ---- store x.  Accumulator: token fr, features 16db + 4fg .. +3 (16 bytes per lane, 64 contiguous bytes per token)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int64_t t = tok0 + qb * 16 + fr;
    if (t < rows) {
      float* xr = x + t * ldx;
#pragma unroll
      for (int db = 0; db < DB; ++db) *reinterpret_cast<f32x4*>(xr + db * 16 + fg * 4) = oacc[qb][db];
      if (x16) {  // the stage's last block also leaves an f16 copy for the FPN's lateral convolution (what lmx_k_cast_f32_f16 made)
        half_t* hr = x16 + t * D;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const f32x4 v = oacc[qb][db];
          *reinterpret_cast<half4_t*>(hr + db * 16 + fg * 4) = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
      }
    }
    // the NEXT block's LayerNorm on the rows this wave just finished (they are all in registers: a token's D features sit in
    // its four lanes): h_n = LayerNorm(x; gam_n, bet_n) in f16, the input of that block's qkv projection — the standalone
    // LayerNorm launch it replaces reads the f32 stream again.  Same arithmetic form as the LayerNorm at the top of this kernel.
    if (h_n) {
      float sm = 0.f;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const f32x4 v = oacc[qb][db];
        sm += (v[0] + v[1]) + (v[2] + v[3]);
      }
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm * (1.0f / D);
      float sq = 0.f;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const f32x4 a = oacc[qb][db] - mean;
        sq += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
      }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = __builtin_amdgcn_rsqf(sq * (1.0f / D) + eps);
      if (t < rows) {
        half_t* hr = h_n + t * D;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const int d = db * 16 + fg * 4;
          const f32x4 g = *reinterpret_cast<const f32x4*>(gam_n + d), c = *reinterpret_cast<const f32x4*>(bet_n + d);
          const f32x4 v = oacc[qb][db];
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)((v[e] - mean) * rstd * g[e] + c[e]);
          *reinterpret_cast<half4_t*>(hr + d) = o;
        }
      }
    }
  }
}

template <int D, int QB, int NW, int NST, int OCC, bool INLN, bool RING = false>
int launch(float* x, int64_t ldx, const half_t* hn, const float* gam, const float* bet, float eps, const half_t* w1, const float* b1,
           const half_t* w2, const float* b2, int64_t rows, half_t* x16, const float* gam_n, const float* bet_n, half_t* h_n, hipStream_t st) {
  constexpr int DP = D <= 128 ? 128 : (D <= 256 ? 256 : 512);
  const size_t smem = (size_t)NST * (HC * DP * 2 + DP * 64) + 4 * D * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_mlp_kernel<D, QB, NW, NST, OCC, INLN, RING>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    attr_set = true;
  }
  const int64_t per = NW * 16 * QB;
  const int64_t nb = (rows + per - 1) / per;
  LMX_REQUIRE(nb < 0x7fffffffll, "lmx_k_ln_mlp: too many rows");
  // staggered LDS-DMA issue (see the kernel): -3 % at D = 224, -6 % at D = 448, neutral at D = 112 (four waves at three workgroups
  // per CU are de-phased anyway); identical bits.  LMX_MLP_STAGGER=0 / 1 overrides.
  static int stagger_env = -2;
  if (stagger_env == -2) stagger_env = getenv("LMX_MLP_STAGGER") ? atoi(getenv("LMX_MLP_STAGGER")) : -1;
  const int stagger = stagger_env >= 0 ? stagger_env : (D >= 224 ? 1 : 0);
  hipLaunchKernelGGL((ln_mlp_kernel<D, QB, NW, NST, OCC, INLN, RING>), dim3((unsigned)nb), dim3(NW * 64), smem, st, x, ldx, hn, gam, bet, eps, w1, b1, w2,
                     b2, rows, x16, gam_n, bet_n, h_n, stagger);
  return lmx_launch_check("ln_mlp_kernel");
}

}  // namespace lmx_mlp
using namespace lmx_mlp;

extern "C" int lmx_k_ln_mlp(float* x, int64_t ldx, const float* gamma, const float* beta, const void* w1, const float* b1,
                            const void* w2, const float* b2, int64_t rows, int D, float eps, void* workspace, void* x16,
                            const float* gamma_next, const float* beta_next, void* h_next, lmx_stream_t stream) {
  LMX_REQUIRE(x && gamma && beta && w1 && b1 && w2 && b2 && workspace, "lmx_k_ln_mlp: null pointer");
  // (D = 448 is a development configuration: instantiated, measured slower than the unfused launches, reachable only with LMX_MLP448 set)
  LMX_REQUIRE(D == 112 || D == 224 || (D == 448 && getenv("LMX_MLP448")), "lmx_k_ln_mlp: D=%d (built for the Hiera stage widths 112 and 224)", D);
  LMX_REQUIRE(rows > 0 && rows < 0x7fffffffll && ldx >= D && ldx % 4 == 0, "lmx_k_ln_mlp: rows=%lld ldx=%lld", (long long)rows,
              (long long)ldx);
  LMX_REQUIRE(aligned16(x) && aligned16(gamma) && aligned16(beta) && aligned16(w1) && aligned16(b1) && aligned16(w2) && aligned16(b2) &&
                  aligned16(workspace),
              "lmx_k_ln_mlp: pointers must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const half_t* W1 = reinterpret_cast<const half_t*>(w1);
  const half_t* W2 = reinterpret_cast<const half_t*>(w2);
  half_t* X16 = reinterpret_cast<half_t*>(x16);
  half_t* HN = reinterpret_cast<half_t*>(h_next);
  LMX_REQUIRE(!h_next || (gamma_next && beta_next && aligned16(gamma_next) && aligned16(beta_next) && ((((uintptr_t)h_next) & 7) == 0)),
              "lmx_k_ln_mlp: h_next needs gamma_next / beta_next (16-byte aligned) and an 8-byte aligned output");
  LMX_REQUIRE(!x16 || ((((uintptr_t)x16) & 7) == 0), "lmx_k_ln_mlp: x16 must be 8-byte aligned");
  static int one_per_cu = -1, split_ln = 0;  // LMX_MLP_ONE_PER_CU=1: the 8-wave, one-workgroup-per-CU configuration of the narrow width too
  if (one_per_cu < 0) {
    one_per_cu = getenv("LMX_MLP_ONE_PER_CU") ? 1 : 0;
    split_ln = getenv("LMX_MLP_SPLIT_LN") ? 1 : 0;  // the round-1 form: a LayerNorm launch into the workspace, then the fused MLP on it
  }
  // D = 112: 4 waves x 32 tokens, 48 KB ring, three workgroups per CU.  D = 224: 8 waves x 32 tokens, 128 KB ring, one per CU.
  // D = 448 (Hiera-B+ stage 3), round 3: two configurations, both correct, neither faster than the unfused launches at 122 880 tokens
  // (LayerNorm + fc1 + fc2 = 0.69 ms; profiles/r03_fused_mlp_d448.txt): LMX_MLP448=2: 4 waves x 32 tokens, one wave per SIMD (336
  // token registers), 2 x 64 KB ring: 1.03 ms; LMX_MLP448=1: 8 waves x 16 tokens, two per SIMD: 0.72 ms, 0.69 ms with staggered
  // issue.  Every 128-token tile streams all 3.2 MB of weights through LDS with ONE 64 KB chunk in flight (the ring holds two), 56
  // chunks each behind a vmcnt(0) + barrier.  Not dispatched by lmx/sam.py.
  if (D == 448) {
    static int v448 = -1;
    if (v448 < 0) v448 = getenv("LMX_MLP448") ? atoi(getenv("LMX_MLP448")) : 1;
    if (v448 == 2) return launch<448, 2, 4, 2, 1, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
    return launch<448, 1, 8, 2, 1, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  }
  // fragment reads through a register ring (RING): D = 224 0.72 -> 0.68 ms per 524 288 tokens, D = 112 0.88 -> 0.87 ms, identical bits
  // (tools/mlp_ab.sh); LMX_MLP_RING=0 is the plain form, 2 the ring with D = 112 at two workgroups per CU (no better)
  static int ring = -1;
  if (ring < 0) ring = getenv("LMX_MLP_RING") ? atoi(getenv("LMX_MLP_RING")) : 1;
  static int cfg = -1;  // development: LMX_MLP_CFG selects experimental tilings (tools/mlp_ab.sh)
  if (cfg < 0) cfg = getenv("LMX_MLP_CFG") ? atoi(getenv("LMX_MLP_CFG")) : 0;
  if (cfg == 3 && D == 224) return launch<224, 1, 8, 2, 2, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (cfg == 3 && D == 112) return launch<112, 1, 4, 2, 4, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (cfg == 4 && D == 224) return launch<224, 1, 8, 3, 1, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (cfg == 4 && D == 112) return launch<112, 1, 8, 2, 2, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (cfg == 5 && D == 224) return launch<224, 1, 16, 4, 1, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (cfg == 5 && D == 112) return launch<112, 1, 16, 4, 1, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (cfg == 6 && D == 224) return launch<224, 1, 8, 2, 2, true, false>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (cfg == 6 && D == 112) return launch<112, 2, 8, 2, 2, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (ring && !split_ln && D != 448) {
    if (D == 112 && ring == 2) return launch<112, 2, 4, 3, 2, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
    if (D == 112) return launch<112, 2, 4, 3, 3, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
    // D = 224: 16 tokens per wave (84 token registers instead of 168), 8 waves, TWO workgroups per CU on 2 x 32 KB rings: four waves
    // per SIMD hide the fragment-read latency that two could not — 0.73 -> 0.63 ms per 524 288 tokens (tools/mlp_ab.sh: cfg 3 against
    // the 32-token form, LMX_MLP_CFG=7); at D = 112 the 32-token form at three workgroups per CU stays ahead (0.88 vs 1.01 ms)
    if (cfg == 7) return launch<224, 2, 8, 4, 1, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
    return launch<224, 1, 8, 2, 2, true, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  }
  if (!split_ln) {
    if (D == 112 && !one_per_cu) return launch<112, 2, 4, 3, 3, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
    if (D == 112) return launch<112, 2, 8, 4, 1, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
    return launch<224, 2, 8, 4, 1, true>(x, ldx, nullptr, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  }
  // 1. LayerNorm (norm.hip; f32 stream -> f16 [rows, D] in the workspace), 2. the fused MLP + residual on it
  const int rc = lmx_k_layernorm(x, LMX_F32, ldx, gamma, beta, workspace, LMX_F16, D, (int)rows, D, eps, LMX_ACT_NONE, stream);
  if (rc) return rc;
  const half_t* hn = reinterpret_cast<const half_t*>(workspace);
  if (D == 112 && !one_per_cu) return launch<112, 2, 4, 3, 3, false>(x, ldx, hn, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  if (D == 112) return launch<112, 2, 8, 4, 1, false>(x, ldx, hn, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
  return launch<224, 2, 8, 4, 1, false>(x, ldx, hn, gamma, beta, eps, W1, b1, W2, b2, rows, X16, gamma_next, beta_next, HN, st);
}
