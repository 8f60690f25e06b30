// nms.hip — K8: ultralytics non_max_suppression as the predictor invokes it under
// services/yolo-pipeline/app/main.py:76 (conf filter -> best class -> sort by score -> class offset ->
// torchvision.ops.nms(iou 0.7) -> first max_det).  Integer/index work must be bit-exact, so:
//   * this file is compiled with -ffp-contract=off (no FMA contraction) and IEEE f32 division;
//   * IoU follows torchvision/csrc/ops/cpu/nms_kernel.cpp (not in tree): areas on the class-offset boxes,
//     inter / (iarea + areas[j] - inter), compared as (double)ovr > iou_threshold;
//   * the sort key is (score, anchor index): descending score, ties broken by the LOWER anchor index (torch's
//     argsort(descending=True) leaves tie order unspecified; this is the stable-sort answer).
//
// Kernels
//   nms_filter_kernel : one thread per anchor: max/argmax over nc class scores (first max wins, as torch.max),
//                       conf test, atomic append of a 64-bit key (score bits << 32 | ~anchor) to the image's list.
//   nms_greedy_kernel : one 1024-thread workgroup per image: bitonic sort of the keys in LDS (<=16384 keys =
//                       128 KB of the 160 KB LDS), gather of the sorted class-offset boxes to the workspace, then
//                       greedy suppression in blocks of 64 sorted candidates: wave 0 resolves the block
//                       (64x64 IoU bitmask per lane, 64-step scalar scan with v_readlane), then all 16 waves
//                       suppress the later candidates against the block's kept boxes.  Stops at max_det.
#include "common.h"

namespace {

constexpr int MAX_A = 16384;

__global__ __launch_bounds__(256) void nms_filter_kernel(const float* __restrict__ pred, int n, int A, int nc,
                                                         float conf, unsigned long long* __restrict__ keys,
                                                         int* __restrict__ counts) {
  const int64_t total = (int64_t)n * A;
  const int row_len = 4 + nc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(i / A);
    const int a = (int)(i - (int64_t)img * A);
    const float* row = pred + i * row_len + 4;
    float best = row[0];
    for (int c = 1; c < nc; ++c) {
      const float v = row[c];
      best = v > best ? v : best;
    }
    if (best > conf) {
      const int slot = atomicAdd(&counts[img], 1);
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a);
      keys[(int64_t)img * A + slot] = key;
    }
  }
}

struct Box {
  float x1, y1, x2, y2, area;
};

__device__ __forceinline__ bool iou_gt(const Box& a, const Box& b, double thr) {
  const float xx1 = a.x1 > b.x1 ? a.x1 : b.x1;
  const float yy1 = a.y1 > b.y1 ? a.y1 : b.y1;
  const float xx2 = a.x2 < b.x2 ? a.x2 : b.x2;
  const float yy2 = a.y2 < b.y2 ? a.y2 : b.y2;
  float w = xx2 - xx1;
  w = w > 0.f ? w : 0.f;
  float h = yy2 - yy1;
  h = h > 0.f ? h : 0.f;
  const float inter = w * h;
  const float ovr = inter / (a.area + b.area - inter);
  return (double)ovr > thr;
}

__global__ __launch_bounds__(1024) void nms_greedy_kernel(const float* __restrict__ pred, int A, int nc, double iou,
                                                          int max_det, float max_wh,
                                                          const unsigned long long* __restrict__ keys_g,
                                                          const int* __restrict__ cand_counts, float* __restrict__ sbox,
                                                          float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                          int* __restrict__ out_cls, int* __restrict__ out_src,
                                                          int* __restrict__ out_counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);  // [npow2]
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  const int count = cand_counts[img];
  const int row_len = 4 + nc;
  const float* P = pred + (int64_t)img * A * row_len;

  int npow2 = 64;
  while (npow2 < count) npow2 <<= 1;
  unsigned char* supp = reinterpret_cast<unsigned char*>(smem + (size_t)npow2 * 8);  // [npow2]
  Box* kept = reinterpret_cast<Box*>(smem + (size_t)npow2 * 9);                      // [64]  (npow2*9 % 16 == 0)
  __shared__ int s_nk;
  __shared__ int s_total;
  __shared__ int s_kept_idx[64];

  for (int i = tid; i < npow2; i += 1024) {
    keys[i] = i < count ? keys_g[(int64_t)img * A + i] : 0ull;
    supp[i] = 0;
  }
  if (tid == 0) s_total = 0;
  __syncthreads();

  // ---- bitonic sort, descending
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- gather the sorted, class-offset boxes (xywh2xyxy then + cls*max_wh, all f32 as ultralytics does)
  float* SB = sbox + (int64_t)img * A * 5;
  for (int i = tid; i < count; i += 1024) {
    const unsigned a = 0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull);
    const float* row = P + (int64_t)a * row_len;
    const float cx = row[0], cy = row[1], w = row[2], h = row[3];
    const float dw = w / 2.f, dh = h / 2.f;
    float best = row[4];
    int bc = 0;
    for (int c = 1; c < nc; ++c) {
      const float v = row[4 + c];
      if (v > best) {
        best = v;
        bc = c;
      }
    }
    const float off = (float)bc * max_wh;
    const float x1 = (cx - dw) + off, y1 = (cy - dh) + off, x2 = (cx + dw) + off, y2 = (cy + dh) + off;
    SB[i * 5 + 0] = x1;
    SB[i * 5 + 1] = y1;
    SB[i * 5 + 2] = x2;
    SB[i * 5 + 3] = y2;
    SB[i * 5 + 4] = (x2 - x1) * (y2 - y1);
  }
  __syncthreads();  // SB is re-read by this workgroup only; writes are visible after the barrier (same CU)

  const int lane = tid & 63;
  const int nblk = (count + 63) / 64;
  for (int blk = 0; blk < nblk; ++blk) {
    const int base = blk * 64;
    if (tid < 64) {
      const int i = base + lane;
      const bool valid = i < count;
      Box me{0.f, 0.f, 0.f, 0.f, 0.f};
      if (valid) me = Box{SB[i * 5 + 0], SB[i * 5 + 1], SB[i * 5 + 2], SB[i * 5 + 3], SB[i * 5 + 4]};
      // lane i: mask of later in-block candidates it would suppress
      unsigned long long mask = 0ull;
      for (int j = 0; j < 64; ++j) {
        Box o;
        o.x1 = __shfl(me.x1, j, 64);
        o.y1 = __shfl(me.y1, j, 64);
        o.x2 = __shfl(me.x2, j, 64);
        o.y2 = __shfl(me.y2, j, 64);
        o.area = __shfl(me.area, j, 64);
        if (j > lane && base + j < count && iou_gt(me, o, iou)) mask |= 1ull << j;
      }
      const bool alive0 = valid && supp[i] == 0;
      unsigned long long alive = __ballot(alive0);
      unsigned long long keepm = 0ull;
      const int room = max_det - s_total;
      int nk = 0;
      for (int j = 0; j < 64 && nk < room; ++j) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)(mask & 0xFFFFFFFFull), j);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(mask >> 32), j);
        if ((alive >> j) & 1ull) {
          keepm |= 1ull << j;
          alive &= ~(((unsigned long long)hi << 32) | lo);
          ++nk;
        }
      }
      if ((keepm >> lane) & 1ull) {
        const int pos = __popcll(keepm & ((1ull << lane) - 1ull));
        kept[pos] = me;
        s_kept_idx[pos] = i;
      }
      if (lane == 0) s_nk = nk;
    }
    __syncthreads();
    const int nk = s_nk;
    const int total_before = s_total;
    // write this block's kept detections
    if (tid < nk) {
      const int i = s_kept_idx[tid];
      const int o = total_before + tid;
      const unsigned long long key = keys[i];
      const unsigned a = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      const float* row = P + (int64_t)a * row_len;
      const float cx = row[0], cy = row[1], w = row[2], h = row[3];
      const float dw = w / 2.f, dh = h / 2.f;
      float best = row[4];
      int bc = 0;
      for (int c = 1; c < nc; ++c) {
        const float v = row[4 + c];
        if (v > best) {
          best = v;
          bc = c;
        }
      }
      float* ob = out_boxes + ((int64_t)img * max_det + o) * 4;
      ob[0] = cx - dw;
      ob[1] = cy - dh;
      ob[2] = cx + dw;
      ob[3] = cy + dh;
      out_scores[(int64_t)img * max_det + o] = best;
      out_cls[(int64_t)img * max_det + o] = bc;
      out_src[(int64_t)img * max_det + o] = (int)a;
    }
    // suppress later candidates against the kept boxes of this block
    if (nk > 0 && total_before + nk < max_det) {
      for (int j = base + 64 + tid; j < count; j += 1024) {
        if (supp[j]) continue;
        const Box o{SB[j * 5 + 0], SB[j * 5 + 1], SB[j * 5 + 2], SB[j * 5 + 3], SB[j * 5 + 4]};
        bool s = false;
        for (int k = 0; k < nk && !s; ++k) s = iou_gt(kept[k], o, iou);
        if (s) supp[j] = 1;
      }
    }
    __syncthreads();
    if (tid == 0) s_total = total_before + nk;
    __syncthreads();
    if (total_before + nk >= max_det) break;
  }
  if (tid == 0) out_counts[img] = s_total;
}

}  // namespace

extern "C" int64_t lmx_nms_workspace_bytes(int n, int A) {
  if (n <= 0 || A <= 0) return 0;
  // counts (padded to 256 B) + keys u64 [n][A] + sorted boxes f32 [n][A][5]
  return 256 + ((int64_t)n * 4 + 255) / 256 * 256 + (int64_t)n * A * 8 + (int64_t)n * A * 5 * 4;
}

extern "C" int lmx_k_nms(const float* pred, int n, int A, int nc, float conf, double iou, int max_det, float max_wh,
                         float* boxes, float* scores, int32_t* cls, int32_t* src, int32_t* counts, void* workspace,
                         lmx_stream_t stream) {
  LMX_REQUIRE(pred && boxes && scores && cls && src && counts && workspace, "lmx_k_nms: null pointer");
  LMX_REQUIRE(n > 0 && A > 0 && A <= MAX_A && nc > 0 && max_det > 0, "lmx_k_nms: n=%d A=%d (<=%d) nc=%d max_det=%d", n,
              A, MAX_A, nc, max_det);
  LMX_REQUIRE((((uintptr_t)workspace) & 255) == 0, "lmx_k_nms: workspace must be 256-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  int* cand_counts = reinterpret_cast<int*>(ws);
  const int64_t cnt_bytes = ((int64_t)n * 4 + 255) / 256 * 256;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + cnt_bytes);
  float* sbox = reinterpret_cast<float*>(ws + cnt_bytes + (int64_t)n * A * 8);
  LMX_HIP(hipMemsetAsync(cand_counts, 0, (size_t)n * 4, st));
  int64_t g = ((int64_t)n * A + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(nms_filter_kernel, dim3((unsigned)g), dim3(256), 0, st, pred, n, A, nc, conf, keys, cand_counts);
  int rc = lmx_launch_check("nms_filter_kernel");
  if (rc) return rc;
  int npow2 = 64;
  while (npow2 < A) npow2 <<= 1;
  const size_t smem = (size_t)npow2 * 9 + 64 * sizeof(Box);
  static bool attr_set = false;
  if (!attr_set) {
    LMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_greedy_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)MAX_A * 9 + 64 * sizeof(Box))));
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_greedy_kernel, dim3(n), dim3(1024), smem, st, pred, A, nc, iou, max_det, max_wh, keys, cand_counts,
                     sbox, boxes, scores, cls, src, counts);
  return lmx_launch_check("nms_greedy_kernel");
}
