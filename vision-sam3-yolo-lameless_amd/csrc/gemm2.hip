// gemm2.hip — the second-generation dense GEMM (a_mode 0): 256(m) x 128(n) x 64 block tile, LDS-DMA staging.
//
// Why a second kernel: in gemm.hip every staged byte crosses the register file (global_load -> VGPR -> ds_write_b128);
// on gfx950 ds_write_b128 moves only ~79 B/clk/CU (MI355X_MICROARCH.md §LDS), so a 128x128x64 k-tile pays ~415 LDS-write
// cycles + ~256 LDS-read cycles against 512 MFMA cycles: the kernel is LDS-bound near 800 TFLOP/s.  Here
//   * staging is `buffer_load_dwordx4 ... lds` (LDS-DMA): no VGPR round trip, no ds_write; the buffer descriptor's range
//     check supplies the zero fill for rows >= M / >= N, a per-lane predicate supplies it for the K tail;
//   * the LDS image keeps the (chunk ^ row&7) swizzle of gemm.hip; because an LDS-DMA wave-instruction writes 64 x 16 B
//     LINEARLY, the swizzle is applied to the per-lane SOURCE address instead (guide §5.4 rule 21): lane L of the
//     instruction that fills rows 8j..8j+7 loads logical chunk (L&7)^(L>>3) of row 8j+(L>>3);
//   * 3-slot LDS ring (3 x 48 KB = 144 KB of the 160 KB), two k-tiles in flight, ONE raw s_barrier per k-tile with a
//     counted s_waitcnt vmcnt (never 0 inside the loop; guide §5 "Pipelining across barriers");
//   * 8 waves (two per SIMD) as 4(m) x 2(n), each owning 64 x 64 = 4 x 4 MFMA 16x16x32 fragments: while one wave of a
//     SIMD waits on its counted vmcnt, issues its 6 LDS-DMAs or reads fragments, its partner keeps the matrix pipe busy
//     (a 4-wave / one-per-SIMD variant of this kernel measured 949 TFLOP/s at 8192^3: DMA issue and fragment reads of a
//     lone wave cannot hide behind its own MFMAs).  Per k-tile the CU reads 128 KB of fragments (512 LDS cycles) against
//     1024 MFMA cycles per SIMD.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// (no anonymous namespace here: hipcc does not emit the host stub of an internal-linkage kernel template that is only
// instantiated from another template — the library then fails to load with an undefined symbol)
#ifdef LMX_DBG_TIMELINE
__device__ unsigned long long lmx_tl[8 * 16384];  // per tile: hw id, xcc id, start, main loop end, end (100 MHz clock), bid
extern "C" int lmx_dbg_get_timeline(void* host, int64_t bytes) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lmx_tl), bytes); }
#define TL(i, v) if (threadIdx.x == 0 && swz < 16384) lmx_tl[swz * 8 + (i)] = (v)
#else
#define TL(i, v)
#endif
namespace lmx_gemm2 {

__device__ __forceinline__ float act_apply(float v, int act) { return lmx_act(v, act); }

// BM x BN x BK tile, NSTAGE-slot LDS ring, (BM/64) x (BN/64) waves of 64 x 64 outputs.
// AMODE 1: A is generated from an NHWC image batch (3x3, pad 1, stride 1|2; gemm.hip's a_mode 1) — requires Cin % BK == 0
// so that a k-tile lies inside one filter tap: the tap (ky,kx) is then wave-uniform per k-tile and a lane only adds a
// constant to its pixel offset; out-of-image taps take the out-of-range offset and the descriptor returns zeros.
// AMODE 2: "pooled rows" — the A rows are a [n][H][W] token grid and the OUTPUT is its 2 x 2 max-pool: GEMM row m reads token
// (image, 2*gy + (m&3)/2, 2*gx + (m&3)%2) of pool group m/4 = (image, gy, gx), so the four tokens of a group are four
// consecutive accumulator rows of one wave; the epilogue takes their maximum (two lane shuffles) and writes ONE row, M/4 in
// all.  Replaces GEMM (f32, 4 B per element written) + maxpool2 (read back, written again) at Hiera's stage transitions
// (TF sam2 Sam2MultiScaleBlock: `do_pool(self.proj(hidden_states))`); max commutes with nothing that is done before it
// here (bias is added first, as the unfused kernels do), so the bits are those of the two-kernel form.
// STAG 1: the two halves of the waves (w and w + NWAVE/2 share a SIMD) run ONE barrier interval apart, and a k-tile is
// two intervals: X = fragment reads + counted vmcnt + next LDS-DMA, Y = the MFMAs.  While one wave of a SIMD is in Y the
// other is in X: the matrix pipe sees MFMAs in every interval instead of every second one (guide: 8-phase template).
// waves per SIMD the register allocation must allow: two workgroups per CU when the ring leaves room for them.  Stated as
// a thread bound (waves x 256 threads = that many waves on each of the 4 SIMDs) rather than as a minimum-occupancy hint: the
// hint makes hipcc schedule up to the cap and spill a few registers in the 128-register kernels.
constexpr int smem_bytes(int BM, int BN, int BK, int NSTAGE) { return NSTAGE * (BM + BN) * BK * 2 + 2 * BN * 4; }  // ring + bias/scale
constexpr int waves_per_simd(int BM, int BN, int BK, int NSTAGE) {
  const int smem = smem_bytes(BM, BN, BK, NSTAGE), nwave = (BM / 64) * (BN / 64);
  const int wps = (smem <= 80 * 1024 ? 2 : 1) * nwave / 4;
  return wps < 1 ? 1 : wps;
}
template <int OUT_DT, int BM, int BN, int BK, int NSTAGE, int AMODE, int STAG>
__global__ __launch_bounds__(waves_per_simd(BM, BN, BK, NSTAGE) * 256) void gemm2_kernel(const lmx_gemm_desc p, const int ntiles, const int nt_ok) {
  constexpr int NWAVE = (BM / 64) * (BN / 64);
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int A_INSTR = BM * BK * 2 / 1024 / NWAVE;  // LDS-DMA wave-instructions (1 KB each) per wave per k-tile
  constexpr int W_INSTR = BN * BK * 2 / 1024 / NWAVE;
  constexpr int ROWS_PER_INSTR = 1024 / (BK * 2);       // 8 (BK=64) or 16 (BK=32)
  constexpr int CHUNKS = BK / 8;                        // 16-byte chunks per row: 8 or 4
  constexpr int LA = NSTAGE - 1;                        // k-tiles issued ahead
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % (BM / 64), wn = wave / (BM / 64);  // (BM/64)(m) x 2(n) waves, 64 x 64 outputs each

  // XCD-aware tile order: the hardware deals workgroups round-robin over the 8 XCDs; XCD x owns the contiguous tile range
  // [base, base + cnt) (n fastest, so the n-tiles of one A row panel share that XCD's L2).  With gridDim.x < ntiles the
  // workgroup is PERSISTENT and walks its XCD's range with the stride of the workgroups resident there: no block launch,
  // no end-of-kernel store drain and no cold prologue between two tiles.
  const int bid = blockIdx.x, xcd = bid & 7;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  const int stride = ((int)gridDim.x + 7 - xcd) >> 3;  // workgroups on this XCD
  const int NT = (p.N + BN - 1) / BN;
  for (int li = bid >> 3; li < cnt; li += stride) {
  const int swz = base + li;
  // split-K (f32 output only; lmx.h): work item swz = tile * S + s computes k-tiles [s nk / S, (s + 1) nk / S) of its tile into
  // the s-th partial output; bias and residual go into partial 0, the consumer adds the partials up in a fixed order
  const int S_ = p.split_k > 1 ? p.split_k : 1;
  const int tile_ = swz / S_, sp_ = swz - tile_ * S_;
  const int mt = tile_ / NT, nt = tile_ - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  TL(0, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));
  TL(1, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));
  TL(2, wall_clock64());
  TL(5, (unsigned long long)bid);

  // buffer descriptors over this block's row panels: the hardware range check returns 0 beyond the last valid byte
  const int hw_out = (AMODE == 1) ? p.Ho * p.Wo : 1;
  const int img0 = (AMODE == 1) ? m0 / hw_out : 0;  // first image touched by this tile
  const char* Ab = reinterpret_cast<const char*>(p.A) +
                   ((AMODE == 1) ? (int64_t)img0 * p.H * p.W_ * p.lda * 2 : (AMODE == 2 ? (int64_t)0 : (int64_t)m0 * p.lda * 2));
  const char* Wb = reinterpret_cast<const char*>(p.W) + (int64_t)n0 * p.K * 2;
  // (a_rep > 1, a_mode 0: A's own K range is K / a_rep columns, walked a_rep times — lmx.h)
  const int Ka = (AMODE == 0 && p.a_rep > 1) ? p.K / p.a_rep : p.K;
  int64_t a_bytes = (AMODE == 1) ? ((int64_t)(p.M / hw_out - img0) * p.H * p.W_ - 1) * p.lda * 2 + (int64_t)p.Cin * 2
                                 : ((int64_t)(p.M - (AMODE == 2 ? 0 : m0) - 1) * p.lda + Ka) * 2;
  int64_t w_bytes = (int64_t)(p.N - n0) * p.K * 2;
  if (a_bytes > 0x7FFFFFF0ll) a_bytes = 0x7FFFFFF0ll;
  if (w_bytes > 0x7FFFFFF0ll) w_bytes = 0x7FFFFFF0ll;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ab), 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wb), 0, (int)w_bytes, 0x00020000);

  // swizzle: physical 16-B slot = logical chunk ^ sw(row).  BK=64 (128-B rows, 2 per bank row): sw = row & 7.
  // BK=32 (64-B rows, 4 per bank row): sw = (-(row >> 2)) & 3.  Either makes each 16-lane ds_read_b128 group hit 16
  // distinct slots of the 256-B bank row.  An LDS-DMA instruction writes its 64 lanes linearly, so the SOURCE is permuted.
  const int lrow = lane / CHUNKS;
  const int lsw = (BK == 64) ? (lrow & 7) : ((-(lrow >> 2)) & 3);
  const int lchunk = (lane % CHUNKS) ^ lsw;
  unsigned a_off[A_INSTR], w_off[W_INSTR];
  int a_yx[A_INSTR];  // AMODE 1: (iy0 << 16) | (ix0 & 0xffff) of the row's top-left tap, -1 for rows >= M
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (wave * A_INSTR + j) * ROWS_PER_INSTR + lrow;
    if (AMODE == 0) {
      a_off[j] = (unsigned)(row * (int)p.lda * 2 + lchunk * 16);
      a_yx[j] = 0;
    } else if (AMODE == 2) {
      const int m = m0 + row, g = m >> 2, sub = m & 3;
      const int Wq = p.W_ >> 1, gpi = (p.H >> 1) * Wq;  // pool groups per image row / per image
      const int img = g / gpi, rem = g - img * gpi;
      const int gy = rem / Wq, gx = rem - gy * Wq;
      const int tok = (img * p.H + 2 * gy + (sub >> 1)) * p.W_ + 2 * gx + (sub & 1);
      a_off[j] = m < p.M ? (unsigned)(tok * (int)p.lda * 2 + lchunk * 16) : 0x80000000u;  // (the whole A is < 2 GB: checked by the caller)
      a_yx[j] = 0;
    } else {
      const int m = m0 + row;
      const int img = m / hw_out, rem = m - img * hw_out;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy0 = oy * p.conv_stride - 1, ix0 = ox * p.conv_stride - 1;
      // offset of the (possibly out-of-image) top-left tap relative to image img0; valid taps always land >= 0
      a_off[j] = (unsigned)((((img - img0) * p.H + iy0) * p.W_ + ix0) * (int)p.lda * 2 + lchunk * 16);
      a_yx[j] = (m < p.M) ? ((iy0 << 16) | (ix0 & 0xffff)) : 0x80008000;
    }
  }
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j)
    w_off[j] = (unsigned)(((wave * W_INSTR + j) * ROWS_PER_INSTR + lrow) * p.K * 2 + lchunk * 16);
  const unsigned OOB = 0x80000000u;
  const int nk = (p.K + BK - 1) / BK;
  // this work item's k-tiles; the part boundaries are multiples of 64 columns whatever BK is, so every tiling sums the same ranges
  const int nk64 = (p.K + 63) / 64;
  const int kt0 = (int)((int64_t)sp_ * nk64 / S_) * (64 / BK);
  const int kte = sp_ + 1 == S_ ? nk : (int)((int64_t)(sp_ + 1) * nk64 / S_) * (64 / BK);
  const int nloc = kte - kt0;
  const int nkA = (AMODE == 0 && p.a_rep > 1) ? nk / p.a_rep : nk;  // k-tiles of A's own K range (Ka % 64 == 0 then: no tail)
  const bool k_tail_lane = (nk - 1) * BK + lchunk * 8 >= p.K;  // this lane's chunk is past K in the last k-tile

  auto issue = [&](int kt, int slot) {
    char* st = smem + slot * STAGE_BYTES;
    const bool kill = (kt == nk - 1) && k_tail_lane;
    const int soff = kt * (BK * 2);
    if (AMODE != 1) {
      // A columns of k-tile kt: kt mod nkA, as two scalar compare-subtracts (a_rep <= 3; nkA == nk without repetition)
      const int kta = kt - (kt >= nkA ? nkA : 0) - (kt >= 2 * nkA ? nkA : 0);
      const int soff_a = (AMODE == 0) ? kta * (BK * 2) : soff;
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        char* dst = st + (wave * A_INSTR + j) * 1024;
        lds_dma16(a_rs, dst, kill ? OOB : a_off[j], soff_a);
      }
    } else {
      const int k0 = kt * BK;          // wave-uniform: the whole k-tile sits in one filter tap
      const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int tap_off = ((ky * p.W_ + kx) * (int)p.lda + ci0) * 2;
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        char* dst = st + (wave * A_INSTR + j) * 1024;
        const int iy = (a_yx[j] >> 16) + ky, ix = (int)(short)(a_yx[j] & 0xffff) + kx;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W_;
        lds_dma16(a_rs, dst, ok ? a_off[j] + (unsigned)tap_off : OOB, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
      char* dst = st + BM * BK * 2 + (wave * W_INSTR + j) * 1024;
      lds_dma16(w_rs, dst, kill ? OOB : w_off[j], soff);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias and LayerScale of this tile's BN channels: requested now, parked in LDS (behind the ring) after the main loop,
  // read per fragment by the epilogue — their latency hides behind the k-loop at the price of two registers
  float* bl = reinterpret_cast<float*>(smem + NSTAGE * STAGE_BYTES);  // [2][BN]
  // (the convolution variant has no registers to spare across its k-loop and asks after it)
  float bias_v = 0.f, scale_v = 1.f;
  if (AMODE != 1 && tid < BN && n0 + tid < p.N) {
    if (p.bias && sp_ == 0) bias_v = p.bias[n0 + tid];
    if (p.scale) scale_v = p.scale[n0 + tid];
  }

#pragma unroll
  for (int t = 0; t < LA; ++t)
    if (t < nloc) issue(kt0 + t, t);

  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (BK == 64) ? (frow & 7) : ((-(frow >> 2)) & 3);
  constexpr int PT = A_INSTR + W_INSTR;
  if constexpr (STAG == 0) {
  for (int kt = 0; kt < nloc; ++kt) {  // (kt: index among this work item's k-tiles; the data is k-tile kt0 + kt)
    // tile kt has landed once only the LDS-DMAs of the (at most LA-1) younger tiles are outstanding
    const int left = nloc - 1 - kt;
    wait_tiles<PT>(left < LA - 1 ? left : LA - 1);
    __builtin_amdgcn_s_barrier();
#ifdef LMX_DBG_TIMELINE
    if (kt == 0) { TL(6, wall_clock64()); }
    if (kt == 1) { TL(7, wall_clock64()); }
#endif
    if (kt + LA < nloc) issue(kt0 + kt + LA, (kt + LA) % NSTAGE);
    const char* st = smem + (kt % NSTAGE) * STAGE_BYTES;
    const half_t* as = reinterpret_cast<const half_t*>(st) + (wm * 64 + frow) * BK;
    const half_t* ws = reinterpret_cast<const half_t*>(st + BM * BK * 2) + (wn * 64 + frow) * BK;
#ifdef LMX_DBG_NOMFMA
    if (p.M < 0)  // development probe: keep the staging pipeline, drop fragment reads and MFMAs
#endif
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int coff = (((ks << 2) + fq) ^ fsw) << 3;
      half8_t af[4], wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const half8_t*>(ws + j * 16 * BK + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const half8_t*>(as + i * 16 * BK + coff);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  } else {
  // ---- staggered two-interval schedule.  Interval numbering: group 0 runs X(kt) in interval 2kt and Y(kt) in 2kt+1,
  // group 1 one interval later.  RAW: tile kt+1 is retired by every wave's counted vmcnt in its X(kt) (intervals 2kt,
  // 2kt+1) and first read in X(kt+1) (intervals >= 2kt+2).  WAR: X(kt) restages the slot of tile kt-1, whose last reads
  // (group 1's X(kt-1), interval 2kt-1, drained by lgkmcnt(0) before its barrier) precede interval 2kt.
  const int grp = wave >= NWAVE / 2 ? 1 : 0;
  {
    const int y = nloc - 1 < LA - 1 ? nloc - 1 : LA - 1;  // tiles 1.. may stay in flight; tile 0 must have landed
    wait_tiles<PT>(y);
  }
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nloc; ++kt) {
    const char* st = smem + (kt % NSTAGE) * STAGE_BYTES;
    const half_t* as = reinterpret_cast<const half_t*>(st) + (wm * 64 + frow) * BK;
    const half_t* ws = reinterpret_cast<const half_t*>(st + BM * BK * 2) + (wn * 64 + frow) * BK;
    half8_t af[BK / 32][4], wf[BK / 32][4];
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int coff = (((ks << 2) + fq) ^ fsw) << 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[ks][j] = *reinterpret_cast<const half8_t*>(ws + j * 16 * BK + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const half8_t*>(as + i * 16 * BK + coff);
    }
    if (kt + 1 < nloc) {
      const int rest = nloc - 2 - kt;  // tiles younger than kt+1 that exist
      wait_tiles<PT>(rest < LA - 2 ? (rest < 0 ? 0 : rest) : (LA - 2 < 0 ? 0 : LA - 2));
    }
    if (kt + LA < nloc) issue(kt0 + kt + LA, (kt + LA) % NSTAGE);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (!grp) __builtin_amdgcn_s_barrier();  // re-align the two groups
  }

  // ---- epilogue through LDS.  In the accumulator a lane owns 4 channels of 16 different rows, so direct stores are
  // 8-byte pieces at a row stride (16 store instructions per wave, 32-byte segments).  Each wave instead transposes its
  // 64 x 64 tile through its own slice of the (now idle) staging LDS and writes whole 128-/256-byte row segments with
  // 16-byte lane stores; the residual is read in the same coalesced shape.  bias / activation / LayerScale are applied
  // in registers on the way in.
  if (AMODE == 1 && tid < BN && n0 + tid < p.N) {
    if (p.bias && sp_ == 0) bias_v = p.bias[n0 + tid];
    if (p.scale) scale_v = p.scale[n0 + tid];
  }
  if (tid < BN) {
    bl[tid] = bias_v;
    bl[BN + tid] = scale_v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a raw s_barrier does not wait for this wave's LDS writes
  __builtin_amdgcn_s_barrier();  // every wave is done reading the last k-tile; bias / scale are in LDS
  TL(3, wall_clock64());
#ifdef LMX_DBG_NOEPI
  if (p.M > 0 && acc[0][0][0] != 12345.f) continue;  // development probe: no epilogue
#endif
  // a large f16 output whose rows are whole 128-byte lines streams past L2 (`nt`): written back normally it evicts the A
  // and W panels the co-resident tiles are re-reading (measured: fc1 of Hiera stage 3 561 -> 628 TFLOP/s, stage 2 387 -> 549;
  // ragged rows and the in-place f32 residual update lose with nt and keep the default policy)
  const bool nt_out = nt_ok && OUT_DT == LMX_F16 && ((p.ldc * 2) & 127) == 0 && (int64_t)p.M * p.N >= (16ll << 20);
  char* my = smem + wave * 4608;  // 32 rows x 144 B (f16) or 16 rows x 272 B (f32) per pass
  // The activation and the presence of a LayerScale vector are compile-time inside the body (one uniform switch per tile
  // instead of one per fragment: the independent activation chains of a pass interleave, no branch separates them).
  auto epilogue = [&](auto act_c, auto scale_c) {
    constexpr int ACT = decltype(act_c)::value;
    constexpr bool SCALE = decltype(scale_c)::value;
    // (bias / scale are re-read from LDS per fragment: held in registers they are 32 live VGPRs on top of the 64 of
    // the accumulators in kernels capped at 128)
    const float* blj = bl + wn * 64 + fq * 4;
    auto finish = [&](f32x4 v, int j) -> f32x4 {
      v += *reinterpret_cast<const f32x4*>(blj + j * 16);
      if constexpr (ACT == LMX_ACT_GELU) {  // two values per v_pk_fma_f32 chain
        const f32x2 g0 = gelu_pk(f32x2{v[0], v[1]}), g1 = gelu_pk(f32x2{v[2], v[3]});
        v = f32x4{g0[0], g0[1], g1[0], g1[1]};
      } else if constexpr (ACT != LMX_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
      }
      if constexpr (SCALE) v *= *reinterpret_cast<const f32x4*>(blj + BN + j * 16);
      return v;
    };
    if constexpr (OUT_DT == LMX_F16 && AMODE == 2) {
      // pooled rows, f16 out (Hiera's pooled queries): as the f32 form below, 8 bytes per lane
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v = finish(acc[pass][j], j);
          // (the unfused path rounds to f16 first and pools the f16 values: rounding is monotonic, so max-then-round is the same)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaxf(v[e], __shfl_xor(v[e], 1, 64));
            v[e] = fmaxf(v[e], __shfl_xor(v[e], 2, 64));
          }
          const int m = m0 + wm * 64 + pass * 16 + frow;
          const int n = n0 + wn * 64 + j * 16 + fq * 4;
          if ((frow & 3) == 0 && m < p.M && n < p.N)
            *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.C) + (int64_t)(m >> 2) * p.ldc + n) =
                half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (OUT_DT == LMX_F16) {
      half_t* t16 = reinterpret_cast<half_t*>(my);
      constexpr int RS = 72;  // halfs per LDS row: 64 + 8 (16-byte pad keeps ds_read_b128 aligned and spreads banks)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = finish(acc[pass * 2 + ii][j], j);
            const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4_t*>(t16 + (ii * 16 + frow) * RS + j * 16 + fq * 4) = o;
          }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + (lane >> 3), c8 = lane & 7;
          half8_t o = *reinterpret_cast<const half8_t*>(t16 + row * RS + c8 * 8);
          const int m = m0 + wm * 64 + pass * 32 + row;
          const int n = n0 + wn * 64 + c8 * 8;
          if (m < p.M && n < p.N) {
            if (p.res) {
              const int mr = p.res_rows > 0 ? m % p.res_rows : m;
              const half8_t rr = *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.res) + (int64_t)mr * p.ldr + n);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rr[e]);
            }
            half8_t* dst = reinterpret_cast<half8_t*>(reinterpret_cast<half_t*>(p.C) + (int64_t)m * p.ldc + n);
            if (nt_out)  // (asm: hipcc merges a __builtin_nontemporal_store with the plain store of the other branch and drops nt)
              asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(o) : "memory");
            else
              *dst = o;
          }
        }
      }
    } else if constexpr (AMODE == 2) {
      // pooled rows: the four tokens of a pool group are accumulator rows 4q .. 4q+3 of a 16-row block = lanes that differ in
      // bits 0-1 of the row index; their maximum goes out as one row, straight from the accumulator layout (16 bytes per lane,
      // 64 contiguous bytes per output row and instruction: a quarter of the rows, so the transposer is not worth its trip)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v = finish(acc[pass][j], j);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaxf(v[e], __shfl_xor(v[e], 1, 64));
            v[e] = fmaxf(v[e], __shfl_xor(v[e], 2, 64));
          }
          const int m = m0 + wm * 64 + pass * 16 + frow;
          const int n = n0 + wn * 64 + j * 16 + fq * 4;
          if ((frow & 3) == 0 && m < p.M && n < p.N) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)(m >> 2) * p.ldc + n) = v;
          __builtin_amdgcn_sched_barrier(0);  // one fragment at a time: interleaved, the sixteen of them spill
        }
    } else {
      float* t32 = reinterpret_cast<float*>(my);
      constexpr int RS = 68;  // floats per LDS row: 64 + 4
      // (requesting the residual rows one or two passes ahead measured 6-15 % SLOWER on the model's f32 shapes: the tile's
      // read-modify-write of the residual stream runs at the memory system's mixed read/write rate with one pass in
      // flight already — all 256 CUs reach their epilogues together and move 8 TB/s between them while it lasts;
      // profiles/r02_gemm_epilogue.txt)
      const int lrow = lane >> 4, c16 = lane & 15;
      const int n = n0 + wn * 64 + c16 * 4;
      const bool has_res = p.res != nullptr && sp_ == 0;
      float* Cs = reinterpret_cast<float*>(p.C) + (int64_t)sp_ * p.split_stride;  // this partial's output
      f32x4 rr[4];
      auto load_res = [&](int pass) {
        f32x4* d = rr;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = m0 + wm * 64 + pass * 16 + it * 4 + lrow;
          d[it] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (has_res && m < p.M && n < p.N) {
            const int mr = p.res_rows > 0 ? m % p.res_rows : m;
            d[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (int64_t)mr * p.ldr + n);
          }
        }
      };
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        load_res(pass);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(t32 + frow * RS + j * 16 + fq * 4) = finish(acc[pass][j], j);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 4 + lrow;
          f32x4 o = *reinterpret_cast<const f32x4*>(t32 + row * RS + c16 * 4);
          const int m = m0 + wm * 64 + pass * 16 + row;
          if (m < p.M && n < p.N) {
            if (has_res) o += rr[it];
            *reinterpret_cast<f32x4*>(Cs + (int64_t)m * p.ldc + n) = o;
          }
        }
      }
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    const int act = p.act;
    if (p.scale) {
      if (act == LMX_ACT_NONE) epilogue(std::integral_constant<int, LMX_ACT_NONE>{}, T{});
      else if (act == LMX_ACT_SILU) epilogue(std::integral_constant<int, LMX_ACT_SILU>{}, T{});
      else if (act == LMX_ACT_GELU) epilogue(std::integral_constant<int, LMX_ACT_GELU>{}, T{});
      else epilogue(std::integral_constant<int, LMX_ACT_RELU>{}, T{});
    } else {
      if (act == LMX_ACT_NONE) epilogue(std::integral_constant<int, LMX_ACT_NONE>{}, F{});
      else if (act == LMX_ACT_SILU) epilogue(std::integral_constant<int, LMX_ACT_SILU>{}, F{});
      else if (act == LMX_ACT_GELU) epilogue(std::integral_constant<int, LMX_ACT_GELU>{}, F{});
      else epilogue(std::integral_constant<int, LMX_ACT_RELU>{}, F{});
    }
  }
  __builtin_amdgcn_s_barrier();  // the staging slices are ring memory: nobody restages it before every wave has read its slice
  TL(4, wall_clock64());
  }  // tile loop
}

template <int BM, int BN, int BK, int NSTAGE, int AMODE, int STAG = 0>
int launch2(const lmx_gemm_desc& d, hipStream_t st) {
  const int MT = (d.M + BM - 1) / BM, NT = (d.N + BN - 1) / BN;
  const size_t smem = smem_bytes(BM, BN, BK, NSTAGE);  // the ring is >= NWAVE * 4608 B of epilogue staging for every variant
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<LMX_F16, BM, BN, BK, NSTAGE, AMODE, STAG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<LMX_F32, BM, BN, BK, NSTAGE, AMODE, STAG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  const int ntiles = MT * NT * (d.split_k > 1 ? d.split_k : 1);
  int grid = ntiles;
  static int persist = -1, nt_ok = 1;  // LMX_GEMM2_PERSIST = workgroups per CU of the persistent grid (0: one per tile)
  if (persist < 0) {
    const char* e = getenv("LMX_GEMM2_PERSIST");
    persist = e ? atoi(e) : 0;
    nt_ok = getenv("LMX_GEMM2_NO_NT") ? 0 : 1;
  }
  if (persist > 0 && grid > 256 * persist) grid = 256 * persist;
  if (d.out_dtype == LMX_F16)
    hipLaunchKernelGGL((gemm2_kernel<LMX_F16, BM, BN, BK, NSTAGE, AMODE, STAG>), dim3(grid), dim3(BM * BN / 64), smem, st, d, ntiles, nt_ok);
  else
    hipLaunchKernelGGL((gemm2_kernel<LMX_F32, BM, BN, BK, NSTAGE, AMODE, STAG>), dim3(grid), dim3(BM * BN / 64), smem, st, d, ntiles, nt_ok);
  return lmx_launch_check("gemm2_kernel");
}

}  // namespace lmx_gemm2
using namespace lmx_gemm2;

// called from lmx_k_gemm (gemm.hip) after validation, for a_mode 0.  LMX_GEMM2_VARIANT picks a tiling for experiments:
//   A: 256x128x64, 3 slots (144 KB), 8 waves, 1 block/CU      B: 128x128x64, 2 slots (64 KB), 4 waves, 2 blocks/CU
//   C: 256x128x32, 3 slots (72 KB), 8 waves, 2 blocks/CU      D: 256x128x32, 2 slots (48 KB), 8 waves, 3 blocks/CU
static int g_variant = -1;  // -1: read LMX_GEMM2_VARIANT on first use; lmx_dbg_set_gemm2_variant overrides it (tools/gemm_sweep.py)
extern "C" void lmx_dbg_set_gemm2_variant(int v) { g_variant = v; }

int lmx_gemm2_launch(const lmx_gemm_desc& d, hipStream_t st) {
  if (g_variant < 0) {
    const char* e = getenv("LMX_GEMM2_VARIANT");
    g_variant = e ? e[0] : 0;
  }
  const int variant = g_variant;
  if (d.a_mode == 2) return launch2<256, 256, 64, 2, 2>(d, st);  // pooled rows (f32 out): one tiling
  if (d.a_mode == 1) {  // 3x3 convolution, Cin % 32 == 0 (checked by the caller)
    static int conv_small = -1;
    if (conv_small < 0) conv_small = getenv("LMX_GEMM2_CONV_SMALL") ? 1 : 0;
    const int64_t t256 = (int64_t)((d.M + 255) / 256) * ((d.N + 255) / 256);
    const double q256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
    if (!conv_small && d.N % 256 == 0 && t256 >= 230 && q256 >= 0.75) return launch2<256, 256, 32, 3, 1, 1>(d, st);
    return launch2<256, 128, 32, 3, 1>(d, st);
  }
  switch (variant) {
    case 'A': return launch2<256, 128, 64, 3, 0>(d, st);
    case 'B': return launch2<128, 128, 64, 2, 0>(d, st);
    case 'C': return launch2<256, 128, 32, 3, 0>(d, st);
    case 'D': return launch2<256, 128, 32, 2, 0>(d, st);
    case 'E': return launch2<256, 256, 32, 3, 0>(d, st);  // 16 waves, 96 KB, 1 block/CU: half the L2->LDS bytes per flop
    case 'F': return launch2<256, 256, 32, 4, 0>(d, st);  // ... 128 KB ring: three k-tiles in flight
    case 'H': return launch2<128, 128, 32, 2, 0>(d, st);  // 4 waves, 32 KB: up to 4 blocks/CU for short-K (HBM-bound) shapes
    case 'I': return launch2<128, 128, 32, 3, 0>(d, st);  // 4 waves, 48 KB: 3 blocks/CU
    case 'S': return launch2<256, 128, 32, 6, 0, 1>(d, st);  // staggered wave groups, 6 x 24 KB ring, 1 block/CU
    case 'T': return launch2<256, 128, 64, 3, 0, 1>(d, st);  // staggered, 3 x 48 KB ring
    case 'U': return launch2<256, 128, 32, 3, 0, 1>(d, st);  // staggered, 3 x 24 KB ring, 2 blocks/CU
    case 'V': return launch2<256, 128, 32, 4, 0, 1>(d, st);  // staggered, 4 x 24 KB ring
    case 'W': return launch2<256, 256, 32, 4, 0, 1>(d, st);  // staggered 256x256, 16 waves, 4 x 32 KB ring
    case 'Y': return launch2<256, 256, 32, 3, 0, 1>(d, st);  // staggered 256x256, 3 x 32 KB ring
    case 'Z': return launch2<256, 256, 64, 2, 0>(d, st);     // 256x256x64, 2 x 64 KB: half the barriers, 128-byte DMA row pieces
    default: {
      // measured on the model shapes (profiles/r01_gemm_variants.txt, r01_gemm_staggered_variants.txt): with K < ~1.8k the
      // per-tile prologue/epilogue dominates and two co-resident workgroups (C) hide it; long-K problems, and problems with
      // no more tiles than CUs, prefer the staggered schedule on the 3 x 48 KB ring (T), one workgroup per CU
      // 256 x 256 tiles (staggered, 16 waves, 3 x 32 KB ring, one workgroup per CU) move a third fewer L2->LDS bytes per
      // flop and win 13-24 % where the tile grid wastes little (profiles/r01_gemm_staggered_variants.txt): N a multiple of
      // 256 (or >= 85 % of its last tile when K >= 896 amortises the waste), at least ~one tile per CU, and a last round of
      // tiles that is not mostly empty
      const int64_t nt256 = (d.N + 255) / 256, tiles256 = (int64_t)((d.M + 255) / 256) * nt256;
      const double nfrac = (double)d.N / (double)(nt256 * 256);
      const double q256 = (double)tiles256 / (double)(((tiles256 + 255) / 256) * 256);
      // The rules below were re-derived with COLD A operands (tools/gemm_sweep.py LMX_SWEEP_COLD=1: buffer sets rotated so that
      // the A operands alone exceed the 256 MB Infinity Cache).  A loop over one buffer set keeps A cache-resident and
      // flatters the 256 x 128 tiling, which re-reads A twice as often: the qkv GEMM of Hiera stage 3 (N = 1344) measures 176 us
      // that way with 256 x 128 and 181 with 256 x 256, but 235 vs 199 us behind the LayerNorm that produces its input, as in
      // the model (tools/gemm_context_probe.py, profiles/r02_gemm_cold_sweep.txt).
      if (d.K >= 448 && tiles256 >= 200 && q256 >= 0.75 && nfrac >= 0.85) {
        // 64-deep k-tiles on a 2 x 64 KB ring (half the barriers, 128-byte DMA row pieces): 4-5 % ahead of the staggered
        // 32-deep schedule at K >= 1792 (profiles/r02_gemm_sweep_interleaved.txt, variant Z) and, since the epilogue resolves
        // the activation per tile, 2-5 % ahead at K = 448 .. 1024 too (fc1 of Hiera stage 3 274 vs 289 us, DINO fc1 260 vs 265;
        // profiles/r02_gemm_epilogue.txt).  Eight waves of 128 x 64 on the same ring (a quarter fewer fragment reads) measured
        // 4-7 % BEHIND sixteen of 64 x 64 on every model shape: the k-loop is not bound by LDS read volume.
        return launch2<256, 256, 64, 2, 0>(d, st);
      }
      // short K, N = 224 .. 1536 filling >= 85 % of its 256-wide tiles (Hiera's 224 / 448 / 672 / 896 / 1344 with K = 112, 224):
      // 256 x 256 tiles move a third fewer L2->LDS bytes per flop than 256 x 128 and win 5-15 % cold; K = 224 prefers the
      // 32-deep 3-slot ring (N = 1344: 502 vs 518 us), K = 112 the 64-deep one (N = 672: 809 vs 841 us)
      if (d.K < 448 && d.N >= 224 && d.N <= 1536 && d.N % 256 != 0 && nfrac >= 0.85 && tiles256 >= 200 && q256 >= 0.75)
        return d.K > 128 ? launch2<256, 256, 32, 3, 0>(d, st) : launch2<256, 256, 64, 2, 0>(d, st);
      const int64_t tiles = (int64_t)((d.M + 255) / 256) * ((d.N + 127) / 128);
      return (d.K >= 1792 || tiles <= 256) ? launch2<256, 128, 64, 3, 0, 1>(d, st) : launch2<256, 128, 32, 3, 0>(d, st);
    }
  }
}
