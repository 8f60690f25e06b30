// NOT BUILT (kept for the record; csrc/Makefile does not list it).  Round-2 experiment: measured 0-4 % ahead of the
// one-tile-per-workgroup kernel in isolation and 0.5-1 % BEHIND it in the seven-stream bench (profiles/r02_gemm_epilogue.txt):
// the next tile's first k-tile, issued under the epilogue, only moves its cost into the epilogue (3.1 -> 4.8 us per tile),
// whose stores share the memory path.
// gemm2p.hip — the PERSISTENT form of gemm2.hip's 256 x 256 x 64 tiling (two 64 KB ring slots, sixteen waves of 64 x 64).
//
// Why: the per-tile time line of that tiling (tools/gemm_timeline_probe.py, profiles/r02_gemm_epilogue.txt) shows, per tile of
// a K = 448 problem, 1.8 us from the workgroup's start to its first k-tile in LDS and ~1 us between one workgroup's end and
// the next one's start on the CU — 2.7 of 17 us in which the matrix pipe has nothing to do.  Here one workgroup per CU stays
// resident, walks its tiles, and issues the NEXT tile's first k-tile (LDS-DMA into slot 0) before it runs the
// current tile's epilogue out of slot 1: when the epilogue's last barrier falls the next k-loop starts on data that is there.
//   * the tile order is static: XCD x owns a contiguous tile range (an A row panel's n-tiles share an L2, as in gemm2.hip)
//     and its 32 workgroups walk it with stride 32.  (Handing tiles out through a counter was written first: a returning
//     atomic per tile costs the k-loop its latency — hipcc waits for it where it is issued — and hiding that would mean
//     holding a not-yet-valid register across the loop behind the compiler's back.)
//   * the epilogue transposer must fit one ring slot: 4 KB per wave, XOR-swizzled instead of padded;
//   * same k order (two 32-deep MFMA steps per k-tile) and the same epilogue rounding sequence as every other tiling:
//     identical bits (tests/test_gpu_kernels.py::test_gemm_variants_agree, tools/gemm_decomp_probe.py).
#include "common.h"
#include <stdlib.h>

#include <type_traits>

#ifdef LMX_DBG_TIMELINE
__device__ unsigned long long lmx_tlp[8 * 16384];  // per tile: hw id, xcc id, start, k-loop end, end, bid, first / second k-tile
extern "C" int lmx_dbg_get_timeline_p(void* host, int64_t bytes) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lmx_tlp), bytes); }
#define TLP(i, v) if (threadIdx.x == 0 && swz < 16384) lmx_tlp[swz * 8 + (i)] = (v)
#else
#define TLP(i, v)
#endif
namespace lmx_gemm2p {

constexpr int BM = 256, BN = 256, BK = 64, NWAVE = 16;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 64 KB
constexpr int A_INSTR = BM * BK * 2 / 1024 / NWAVE, W_INSTR = BN * BK * 2 / 1024 / NWAVE;  // 2 + 2 LDS-DMA instructions per wave and k-tile
constexpr int SMEM = 2 * STAGE_BYTES + 2 * BN * 4;  // ring + bias / scale

__device__ __forceinline__ float act_apply(float v, int act) { return lmx_act(v, act); }

template <int OUT_DT>
__global__ __launch_bounds__(1024) void gemm2p_kernel(const lmx_gemm_desc p, const int ntiles, const int nt_ok) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bl = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);                  // [2][BN] bias, scale of the current tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;

  // this workgroup's XCD owns the contiguous tile range [base, base + cnt) (n fastest); its workgroups walk it with their stride
  const int xcd = blockIdx.x & 7;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  const int stride = ((int)gridDim.x + 7 - xcd) >> 3;  // workgroups on this XCD
  const int NT = (p.N + BN - 1) / BN;

  // lane constants of the LDS-DMA plan (tile independent: offsets are relative to the tile's descriptor base)
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lrow & 7);
  unsigned a_off[A_INSTR], w_off[W_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) a_off[j] = (unsigned)(((wave * A_INSTR + j) * 8 + lrow) * (int)p.lda * 2 + lchunk * 16);
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) w_off[j] = (unsigned)(((wave * W_INSTR + j) * 8 + lrow) * p.K * 2 + lchunk * 16);
  const unsigned OOB = 0x80000000u;
  const int nk = (p.K + BK - 1) / BK;
  const bool k_tail_lane = (nk - 1) * BK + lchunk * 8 >= p.K;
  const int frow = lane & 15, fq = lane >> 4, fsw = frow & 7;

  struct Tile {
    int m0, n0;
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
  };
  auto setup = [&](int swz) -> Tile {
    Tile t;
    const int mt = swz / NT, nt = swz - mt * NT;
    t.m0 = mt * BM;
    t.n0 = nt * BN;
    int64_t a_bytes = ((int64_t)(p.M - t.m0 - 1) * p.lda + p.K) * 2, w_bytes = (int64_t)(p.N - t.n0) * p.K * 2;
    if (a_bytes > 0x7FFFFFF0ll) a_bytes = 0x7FFFFFF0ll;
    if (w_bytes > 0x7FFFFFF0ll) w_bytes = 0x7FFFFFF0ll;
    t.a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A) + (int64_t)t.m0 * p.lda * 2), 0, (int)a_bytes, 0x00020000);
    t.w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.W) + (int64_t)t.n0 * p.K * 2), 0, (int)w_bytes, 0x00020000);
    return t;
  };
  auto issue = [&](const Tile& t, int kt, int sl) {
    char* st = smem + sl * STAGE_BYTES;
    const bool kill = (kt == nk - 1) && k_tail_lane;
    const int soff = kt * (BK * 2);
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) lds_dma16(t.a_rs, st + (wave * A_INSTR + j) * 1024, kill ? OOB : a_off[j], soff);
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) lds_dma16(t.w_rs, st + BM * BK * 2 + (wave * W_INSTR + j) * 1024, kill ? OOB : w_off[j], soff);
  };
  int li = blockIdx.x >> 3;
  if (li >= cnt) return;
  Tile cur = setup(base + li);
  issue(cur, 0, 0);
  // bias / LayerScale of a tile are requested one tile ahead (with its first k-tile): asked for at the top of the tile they
  // would be what the first k-tile's wait waits for
  float bias_v = 0.f, scale_v = 1.f;
  if (tid < BN && cur.n0 + tid < p.N) {
    if (p.bias) bias_v = p.bias[cur.n0 + tid];
    if (p.scale) scale_v = p.scale[cur.n0 + tid];
  }

  for (;;) {
    const int m0 = cur.m0, n0 = cur.n0;
#ifdef LMX_DBG_TIMELINE
    const int swz = base + li;
#endif
    TLP(0, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));
    TLP(1, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));
    TLP(2, wall_clock64());
    TLP(5, (unsigned long long)blockIdx.x);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      // one k-tile in flight: everything this wave has outstanding (the k-tile's LDS-DMAs, at kt = 0 also the previous
      // epilogue's stores and the bias loads) has to be in before the barrier
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
#ifdef LMX_DBG_TIMELINE
      if (kt == 0) { TLP(6, wall_clock64()); }
      if (kt == 1) { TLP(7, wall_clock64()); }
#endif
      if (kt + 1 < nk) issue(cur, kt + 1, (kt + 1) & 1);
      const char* st = smem + (kt & 1) * STAGE_BYTES;
      const half_t* as = reinterpret_cast<const half_t*>(st) + (wm * 64 + frow) * BK;
      const half_t* ws = reinterpret_cast<const half_t*>(st + BM * BK * 2) + (wn * 64 + frow) * BK;
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
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }

    if (tid < BN) {
      bl[tid] = bias_v;
      bl[BN + tid] = scale_v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a raw s_barrier does not wait for this wave's LDS writes
    __builtin_amdgcn_s_barrier();                       // every wave is done reading the last k-tile: both slots are free
    TLP(3, wall_clock64());
    li += stride;
    const bool more = li < cnt;
    Tile nx = cur;
    if (more) {
      nx = setup(base + li);
      issue(nx, 0, 0);  // lands in slot 0 while the epilogue below works out of slot 1
      bias_v = 0.f;
      scale_v = 1.f;
      if (tid < BN && nx.n0 + tid < p.N) {  // (bl already holds the current tile's values)
        if (p.bias) bias_v = p.bias[nx.n0 + tid];
        if (p.scale) scale_v = p.scale[nx.n0 + tid];
      }
    }

    // ---- epilogue (gemm2.hip's, with the transposer in slot 1: 4 KB per wave, swizzled)
    const bool nt_out = nt_ok && OUT_DT == LMX_F16 && ((p.ldc * 2) & 127) == 0 && (int64_t)p.M * p.N >= (16ll << 20);
    char* my = smem + STAGE_BYTES + wave * 4096;
    auto epilogue = [&](auto act_c, auto scale_c) {
      constexpr int ACT = decltype(act_c)::value;
      constexpr bool SCALE = decltype(scale_c)::value;
      const float* blj = bl + wn * 64 + fq * 4;
      auto finish = [&](f32x4 v, int j) -> f32x4 {
        v += *reinterpret_cast<const f32x4*>(blj + j * 16);
        if constexpr (ACT == LMX_ACT_GELU) {
          const f32x2 g0 = gelu_pk(f32x2{v[0], v[1]}), g1 = gelu_pk(f32x2{v[2], v[3]});
          v = f32x4{g0[0], g0[1], g1[0], g1[1]};
        } else if constexpr (ACT != LMX_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
        }
        if constexpr (SCALE) v *= *reinterpret_cast<const f32x4*>(blj + BN + j * 16);
        return v;
      };
      if constexpr (OUT_DT == LMX_F16) {
        // 32 rows x 128 B: the 16-byte chunk c of row r sits at chunk c ^ (r & 7) ^ ((r >> 3) & 1)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x4 v = finish(acc[pass * 2 + ii][j], j);
              const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              const int rr = ii * 16 + frow;
              *reinterpret_cast<half4_t*>(my + rr * 128 + (((j * 2 + (fq >> 1)) ^ (rr & 7) ^ ((rr >> 3) & 1)) << 4) + (fq & 1) * 8) = o;
            }
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), c8 = lane & 7;
            half8_t o = *reinterpret_cast<const half8_t*>(my + row * 128 + ((c8 ^ (row & 7) ^ ((row >> 3) & 1)) << 4));
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
              if (nt_out)
                asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(o) : "memory");
              else
                *dst = o;
            }
          }
        }
      } else {
        // 16 rows x 256 B: chunk c of row r sits at chunk c ^ r
        const int lr = lane >> 4, c16 = lane & 15;
        const int n = n0 + wn * 64 + c16 * 4;
        const bool has_res = p.res != nullptr;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          f32x4 rr[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int m = m0 + wm * 64 + pass * 16 + it * 4 + lr;
            rr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has_res && m < p.M && n < p.N) {
              const int mr = p.res_rows > 0 ? m % p.res_rows : m;
              rr[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (int64_t)mr * p.ldr + n);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(my + frow * 256 + (((j * 4 + fq) ^ frow) << 4)) = finish(acc[pass][j], j);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = it * 4 + lr;
            f32x4 o = *reinterpret_cast<const f32x4*>(my + row * 256 + ((c16 ^ row) << 4));
            const int m = m0 + wm * 64 + pass * 16 + row;
            if (m < p.M && n < p.N) {
              if (has_res) o += rr[it];
              *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = o;
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
    __builtin_amdgcn_s_barrier();  // slot 1 is ring memory again: nobody restages it before every wave has read its slice
    TLP(4, wall_clock64());
    if (!more) break;
    cur = nx;
  }
}

}  // namespace lmx_gemm2p
using namespace lmx_gemm2p;

// called from lmx_gemm2_launch for shapes it gives the 256 x 256 x 64 tiling; returns -1 when this form does not apply
// (the caller then launches the one-tile-per-workgroup kernel)
int lmx_gemm2p_launch(const lmx_gemm_desc& d, hipStream_t st) {
  const int MT = (d.M + BM - 1) / BM, NT = (d.N + BN - 1) / BN;
  const int64_t ntiles = (int64_t)MT * NT;
  if (d.a_mode != 0 || ntiles < 2 * 256 || ntiles > 0x3fffffff) return -1;  // fewer than two tiles per CU: nothing to overlap
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2p_kernel<LMX_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2p_kernel<LMX_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set = true;
  }
  static int nt_ok = -1;
  if (nt_ok < 0) nt_ok = getenv("LMX_GEMM2_NO_NT") ? 0 : 1;
  if (d.out_dtype == LMX_F16)
    hipLaunchKernelGGL((gemm2p_kernel<LMX_F16>), dim3(256), dim3(1024), SMEM, st, d, (int)ntiles, nt_ok);
  else
    hipLaunchKernelGGL((gemm2p_kernel<LMX_F32>), dim3(256), dim3(1024), SMEM, st, d, (int)ntiles, nt_ok);
  return lmx_launch_check("gemm2p_kernel");
}
