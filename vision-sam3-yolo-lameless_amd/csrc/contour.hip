// contour.hip — the contour features of services/sam3-pipeline/app/main.py:102-145 (cv2.findContours(RETR_EXTERNAL,
// CHAIN_APPROX_SIMPLE) -> largest contour by cv2.contourArea -> cv2.arcLength / cv2.boundingRect) ON THE DEVICE, for a
// batch of 1080p masks at once: the mask never leaves HBM, 64 bytes per frame do.
//
// host_mask.cpp follows each outer border pixel by pixel (Suzuki-Abe), which is sequential.  The same numbers have a
// parallel form.  Walk the boundary between an 8-connected foreground component C and the OUTSIDE background O (background
// 4-connected to the image frame) crack by crack — a crack = (foreground pixel c, side s) whose neighbour c + s is outside
// — with the foreground on the right.  At the end vertex of a crack, with A = the pixel ahead on the background side and
// B = the pixel straight ahead of c:   A foreground -> turn left,  the next crack belongs to A  (a DIAGONAL step c -> A);
// else B foreground -> go straight, the next crack belongs to B (a UNIT step c -> B);   else turn right around c (no step).
// The pixel sequence of that crack cycle is exactly the border Suzuki-Abe follows (pixels met twice on one-pixel-wide parts
// included), and every step is decided by the 2 x 2 neighbourhood of ONE crack.  So, per component,
//     2 * contourArea = | sum over cracks of (c.x * n.y - n.x * c.y) |,   arcLength = #unit + #diagonal * sqrt(2)
// are plain sums over cracks, exact in integers, in any order.  What is needed besides: labels (which component a pixel
// belongs to; which background is outside) — a union-find connected-component labelling whose root is the component's
// first pixel in raster order, which is also the contour order cv2.findContours' max(..., key=contourArea) breaks ties by.
//
// Kernels: row runs (block max-scan) -> unions between rows (lock-free union-find, min index wins) -> compression ->
// crack sums (wave-reduced when a wave's cracks share a root, as they do on one big mask) -> per-frame selection.
// Checked bit for bit against host_mask.cpp on analytic shapes, random blobs with salt-and-pepper and SAM masks
// (tests/test_gpu_contour.py); cv2 itself is absent, so like the host version: PARITY UNPINNED against real OpenCV.
#include "common.h"

namespace {

__device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(const int* L, int i) {
  int p = ld(L + i);
  while (p != i) {
    i = p;
    p = ld(L + i);
  }
  return i;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a > b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(L + b, a);  // the larger root points to the smaller index
    if (old == b) return;
    b = old;  // somebody re-parented b meanwhile: continue from its new parent
  }
}

// one workgroup per image row: L[i] = index of the first pixel of i's horizontal run of equal type (foreground / background)
__global__ __launch_bounds__(256) void cc_rows_kernel(const uint8_t* __restrict__ mask, int* __restrict__ L, int h, int w, int64_t npix) {
  __shared__ int part[256];
  const int row = blockIdx.x % h, img = blockIdx.x / h;
  const uint8_t* m = mask + (int64_t)img * npix + (int64_t)row * w;
  int* Lr = L + (int64_t)img * (npix + 1) + (int64_t)row * w;
  const int P = (w + 255) / 256;
  const int x0 = threadIdx.x * P;
  int last = -1;  // last run boundary inside this thread's segment
  for (int x = x0; x < x0 + P && x < w; ++x)
    if (x == 0 || (m[x] != 0) != (m[x - 1] != 0)) last = x;
  part[threadIdx.x] = last;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // inclusive max-scan
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : -1;
    __syncthreads();
    if (v > part[threadIdx.x]) part[threadIdx.x] = v;
    __syncthreads();
  }
  int run = threadIdx.x > 0 ? part[threadIdx.x - 1] : -1;
  for (int x = x0; x < x0 + P && x < w; ++x) {
    if (x == 0 || (m[x] != 0) != (m[x - 1] != 0)) run = x;
    Lr[x] = row * w + run;
  }
  if (blockIdx.x % h == 0 && threadIdx.x == 0) L[(int64_t)img * (npix + 1) + npix] = (int)npix;  // the virtual outside node
}

// the same for w % 4 == 0 with word loads and 16-byte label stores: the byte-per-lane form above issues one vector-memory
// instruction per pixel and is bound by that, not by bytes (a wave-instruction moves 64 bytes)
__global__ __launch_bounds__(256) void cc_rows4_kernel(const uint8_t* __restrict__ mask, int* __restrict__ L, int h, int w, int64_t npix) {
  __shared__ int part[256];
  const int row = blockIdx.x % h, img = blockIdx.x / h;
  const uint8_t* m = mask + (int64_t)img * npix + (int64_t)row * w;
  int* Lr = L + (int64_t)img * (npix + 1) + (int64_t)row * w;
  const int P = (((w + 255) / 256) + 3) & ~3;  // pixels per thread, a multiple of 4
  const int x0 = threadIdx.x * P;
  int last = -1;
  bool prev = x0 > 0 && x0 <= w ? m[x0 - 1] != 0 : false;
  const bool prev0 = prev;
  for (int x = x0; x < x0 + P && x < w; x += 4) {
    const unsigned v = *reinterpret_cast<const unsigned*>(m + x);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool f = ((v >> (8 * j)) & 0xffu) != 0;
      if (x + j == 0 || f != prev) last = x + j;
      prev = f;
    }
  }
  part[threadIdx.x] = last;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // inclusive max-scan
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : -1;
    __syncthreads();
    if (v > part[threadIdx.x]) part[threadIdx.x] = v;
    __syncthreads();
  }
  int run = threadIdx.x > 0 ? part[threadIdx.x - 1] : -1;
  prev = prev0;
  // (labels of an image start at img * (npix + 1) ints: 16-byte alignment of a row's labels is not given, so the four labels
  // of a word go out as four dwords; consecutive lanes still cover consecutive 16-byte pieces)
  for (int x = x0; x < x0 + P && x < w; x += 4) {
    const unsigned v = *reinterpret_cast<const unsigned*>(m + x);
    int lab[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool f = ((v >> (8 * j)) & 0xffu) != 0;
      if (x + j == 0 || f != prev) run = x + j;
      prev = f;
      lab[j] = row * w + run;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Lr[x + j] = lab[j];
  }
  if (blockIdx.x % h == 0 && threadIdx.x == 0) L[(int64_t)img * (npix + 1) + npix] = (int)npix;  // the virtual outside node
}

// unions between a row and the row above: foreground 8-connected (N, NW, NE), background 4-connected (N); background on
// the image frame joins the virtual outside node
__global__ __launch_bounds__(256) void cc_union_kernel(const uint8_t* __restrict__ mask, int* __restrict__ L, int n, int h, int w, int64_t npix) {
  const int64_t total = (int64_t)n * npix;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(g / npix);
    const int i = (int)(g - (int64_t)img * npix);
    const int y = i / w, x = i - y * w;
    const uint8_t* m = mask + (int64_t)img * npix;
    int* Li = L + (int64_t)img * (npix + 1);
    const bool fg = m[i] != 0;
    if (!fg && (x == 0 || y == 0 || x == w - 1 || y == h - 1)) uf_union(Li, i, (int)npix);
    if (y == 0) continue;
    const int up = i - w;
    if ((m[up] != 0) == fg) {
      // runs already joined horizontally: a vertical union is only new where the pair (i, up) starts a common stretch
      if (x == 0 || (m[i - 1] != 0) != fg || (m[up - 1] != 0) != fg) uf_union(Li, i, up);
    }
    if (fg) {
      if (x > 0 && m[up - 1] != 0 && m[up] == 0 && m[i - 1] == 0) uf_union(Li, i, up - 1);          // NW, not already implied
      if (x + 1 < w && m[up + 1] != 0 && m[up] == 0 && m[i + 1] == 0) uf_union(Li, i, up + 1);      // NE, not already implied
    }
  }
}

// the same unions for w % 4 == 0, four pixels per thread from two word loads (this row, the row above) and the four bytes
// beside them: a quarter of the vector-memory instructions.  Union-find with min-index roots gives the same labels whatever
// the order of the unions.
__global__ __launch_bounds__(256) void cc_union4_kernel(const uint8_t* __restrict__ mask, int* __restrict__ L, int h, int w, int64_t npix) {
  const int y = blockIdx.x, img = blockIdx.y;
  const uint8_t* m = mask + (int64_t)img * npix + (int64_t)y * w;
  int* Li = L + (int64_t)img * (npix + 1);
  const bool edge_row = y == 0 || y == h - 1;
  for (int x0 = threadIdx.x * 4; x0 < w; x0 += 1024) {
    const unsigned cw = *reinterpret_cast<const unsigned*>(m + x0);
    const unsigned uw = y > 0 ? *reinterpret_cast<const unsigned*>(m - w + x0) : 0u;
    bool c[6], u[6];  // index j + 1 for pixel x0 + j, j = -1 .. 4
    c[0] = x0 > 0 && m[x0 - 1] != 0;
    c[5] = x0 + 4 < w && m[x0 + 4] != 0;
    u[0] = y > 0 && x0 > 0 && m[x0 - 1 - w] != 0;
    u[5] = y > 0 && x0 + 4 < w && m[x0 + 4 - w] != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c[j + 1] = ((cw >> (8 * j)) & 0xffu) != 0;
      u[j + 1] = ((uw >> (8 * j)) & 0xffu) != 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = x0 + j, i = y * w + x;
      const bool fg = c[j + 1];
      if (!fg && (edge_row || x == 0 || x == w - 1)) uf_union(Li, i, (int)npix);
      if (y == 0) continue;
      const int up = i - w;
      if (u[j + 1] == fg) {
        if (x == 0 || c[j] != fg || u[j] != fg) uf_union(Li, i, up);
      }
      if (fg) {
        if (x > 0 && u[j] && !u[j + 1] && !c[j]) uf_union(Li, i, up - 1);
        if (x + 1 < w && u[j + 2] && !u[j + 1] && !c[j + 2]) uf_union(Li, i, up + 1);
      }
    }
  }
}

struct Acc {
  long long* area2;           // [n][npix] sum of cross terms, indexed by root pixel (initialised at roots only)
  unsigned long long* steps;  // [n][npix] unit steps in the low 32 bits, diagonal steps in the high 32
  int* box;                   // [n][npix][4] min x, min y, max x, max y of the component's pixels that own a crack
};

// path compression; a root (the first pixel of a component) also clears its accumulators — the arrays are indexed by root
// pixel and only ever touched there, so nothing else of their 32 bytes per pixel needs initialising
__global__ __launch_bounds__(256) void cc_compress_kernel(int* __restrict__ L, Acc acc, int n, int64_t npix) {
  const int64_t total = (int64_t)n * (npix + 1);
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(g / (npix + 1));
    const int i = (int)(g - (int64_t)img * (npix + 1));
    int* Li = L + (int64_t)img * (npix + 1);
    const int r = uf_find(Li, i);
    if (r != i) {
      __hip_atomic_store(Li + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // only ever shortens a path
    } else if (i < npix) {
      const int64_t k = (int64_t)img * npix + i;
      acc.area2[k] = 0;
      acc.steps[k] = 0;
      acc.box[k * 4 + 0] = 0x7fffffff;
      acc.box[k * 4 + 1] = 0x7fffffff;
      acc.box[k * 4 + 2] = -1;
      acc.box[k * 4 + 3] = -1;
    }
  }
}

// crack sums.  A thread owns one foreground pixel and its (up to four) cracks.
__global__ __launch_bounds__(256) void contour_sum_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ L, Acc acc, int n, int h,
                                                          int w, int64_t npix) {
  const int64_t total = (int64_t)n * npix;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (int64_t base = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 6; base < total; base += nwaves << 6) {
    const int64_t g = base + lane;
    long long a2 = 0;
    unsigned long long st = 0;
    int root = -1, img = 0, px = 0, py = 0;
    if (g < total) {
      img = (int)(g / npix);
      const int i = (int)(g - (int64_t)img * npix);
      const uint8_t* m = mask + (int64_t)img * npix;
      const int* Li = L + (int64_t)img * (npix + 1);
      if (m[i] != 0) {
        const int y = i / w, x = i - y * w;
        const int outside = Li[npix];
        auto is_fg = [&](int px, int py) { return px >= 0 && py >= 0 && px < w && py < h && m[py * w + px] != 0; };
        auto is_out = [&](int px, int py) {
          if (px < 0 || py < 0 || px >= w || py >= h) return true;
          const int q = py * w + px;
          return m[q] == 0 && Li[q] == outside;
        };
        // headings E, S, W, N of the cracks on the N, E, S, W sides; left (background side) = (hy, -hx) with y down
        const int HX[4] = {1, 0, -1, 0}, HY[4] = {0, 1, 0, -1};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int hx = HX[d], hy = HY[d], lx = hy, ly = -hx;
          if (!is_out(x + lx, y + ly)) continue;
          int nx = x, ny = y;
          if (is_fg(x + hx + lx, y + hy + ly)) {  // A: ahead on the background side -> turn left, diagonal step
            nx = x + hx + lx;
            ny = y + hy + ly;
            st += 1ull << 32;
          } else if (is_fg(x + hx, y + hy)) {     // B: straight ahead -> unit step
            nx = x + hx;
            ny = y + hy;
            st += 1ull;
          }
          a2 += (long long)x * ny - (long long)nx * y;
          root = Li[i];
          px = x;
          py = y;
        }
      }
    }
    // most waves have no crack at all; a wave on the border of one big mask has cracks of ONE component: reduce, one atomic
    const bool has = root >= 0;
    const unsigned long long bal = __ballot(has);
    if (bal == 0) continue;
    const int first = __ffsll((long long)bal) - 1;
    const int root0 = __shfl(root, first, 64);
    const int img0 = __shfl(img, first, 64);
    // the lanes that share the first contributor's component reduce together (one atomic pair); the others — cracks of
    // isolated noise pixels and small neighbours — add on their own, to addresses of their own
    const bool grp = has && root == root0 && img == img0;
    long long s2 = grp ? a2 : 0;
    unsigned long long ss = grp ? st : 0;
    int mnx = grp ? px : 0x7fffffff, mny = grp ? py : 0x7fffffff, mxx = grp ? px : -1, mxy = grp ? py : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s2 += __shfl_xor(s2, o, 64);
      ss += __shfl_xor(ss, o, 64);
      mnx = min(mnx, __shfl_xor(mnx, o, 64));
      mny = min(mny, __shfl_xor(mny, o, 64));
      mxx = max(mxx, __shfl_xor(mxx, o, 64));
      mxy = max(mxy, __shfl_xor(mxy, o, 64));
    }
    int64_t k = -1;
    if (lane == first) {
      k = (int64_t)img0 * npix + root0;
    } else if (has && !grp) {
      k = (int64_t)img * npix + root;
      s2 = a2;
      ss = st;
      mnx = mxx = px;
      mny = mxy = py;
    }
    if (k >= 0) {
      atomicAdd(reinterpret_cast<unsigned long long*>(acc.area2 + k), (unsigned long long)s2);
      atomicAdd(acc.steps + k, ss);
      // the box only grows: an atomic that cannot grow it is skipped after a plain (possibly stale = less grown) look
      int* bb = acc.box + k * 4;
      const volatile int* vb = bb;
      if (mnx < vb[0]) atomicMin(bb + 0, mnx);
      if (mny < vb[1]) atomicMin(bb + 1, mny);
      if (mxx > vb[2]) atomicMax(bb + 2, mxx);
      if (mxy > vb[3]) atomicMax(bb + 3, mxy);
    }
  }
}

// per-frame scratch: best = max over external roots of (|area2| << 21 | (0x1fffff - root)) -> largest area, first in raster
// order among equals; count = number of external contours
struct Sel {
  unsigned long long* best;  // [n]
  int* count;                // [n]
};

// every root of an EXTERNAL component (its west neighbour is outside background or the frame) bids for its frame
__global__ __launch_bounds__(256) void contour_select_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ L, Acc acc, Sel sel,
                                                             int n, int w, int64_t npix) {
  const int64_t total = (int64_t)n * npix;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(g / npix);
    const int i = (int)(g - (int64_t)img * npix);
    const uint8_t* m = mask + (int64_t)img * npix;
    if (m[i] == 0) continue;
    const int* Li = L + (int64_t)img * (npix + 1);
    if (Li[i] != i) continue;  // roots only: the first pixel of a component in raster order
    const int x = i % w;
    if (!(x == 0 || (m[i - 1] == 0 && Li[i - 1] == Li[npix]))) continue;
    long long a = acc.area2[(int64_t)img * npix + i];
    a = a < 0 ? -a : a;
    atomicMax(sel.best + img, ((unsigned long long)a << 21) | (unsigned long long)(0x1fffff - i));
    atomicAdd(sel.count + img, 1);
  }
}

__global__ void contour_sel_init_kernel(Sel sel, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sel.best[i] = 0;
  sel.count[i] = 0;
}

// out[8] = area2 (>= 0), unit steps, diagonal steps, min x, min y, max x, max y, number of external contours
__global__ void contour_final_kernel(Acc acc, Sel sel, long long* __restrict__ out, int n, int64_t npix) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= n) return;
  long long* o = out + (int64_t)img * 8;
  if (sel.count[img] == 0) {
    for (int k = 0; k < 8; ++k) o[k] = 0;
    return;
  }
  const unsigned long long key = sel.best[img];
  const int r = 0x1fffff - (int)(key & 0x1fffffull);
  const unsigned long long st = acc.steps[(int64_t)img * npix + r];
  o[0] = (long long)(key >> 21);
  o[1] = (long long)(st & 0xffffffffull);
  o[2] = (long long)(st >> 32);
  const int* bb = acc.box + ((int64_t)img * npix + r) * 4;
  o[3] = bb[0];
  o[4] = bb[1];
  o[5] = bb[2];
  o[6] = bb[3];
  o[7] = sel.count[img];
}

inline unsigned grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  if (b > 256 * 32) b = 256 * 32;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

// labels (npix + 1 ints) + area2 (8 B) + steps (8 B) + box (16 B) per pixel and frame, + 16 B of selection scratch per frame
extern "C" int64_t lmx_contour_workspace_bytes(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  const int64_t npix = (int64_t)h * w;
  return (int64_t)n * ((npix + 1) * 4 + npix * 32 + 16) + 512;
}

extern "C" int lmx_k_contour_features(const uint8_t* mask, int n, int h, int w, int64_t* out, void* workspace, lmx_stream_t stream) {
  LMX_REQUIRE(mask && out && workspace, "lmx_k_contour_features: null pointer");
  // (the selection key packs the root index into 21 bits: 1080p = 2 073 600 pixels < 2^21)
  LMX_REQUIRE(n > 0 && h > 1 && w > 1 && (int64_t)h * w <= 0x1fffff && (int64_t)n * h < 0x7fffffffll && n <= 65535,
              "lmx_k_contour_features: n=%d h=%d w=%d (at most 2^21 - 1 pixels per mask)", n, h, w);
  LMX_REQUIRE(aligned16(workspace), "lmx_k_contour_features: workspace alignment");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t npix = (int64_t)h * w;
  char* ws = reinterpret_cast<char*>(workspace);
  int* L = reinterpret_cast<int*>(ws);
  int64_t off = (((int64_t)n * (npix + 1) * 4) + 15) & ~15ll;
  Acc acc;
  acc.area2 = reinterpret_cast<long long*>(ws + off);
  off += (int64_t)n * npix * 8;
  acc.steps = reinterpret_cast<unsigned long long*>(ws + off);
  off += (int64_t)n * npix * 8;
  acc.box = reinterpret_cast<int*>(ws + off);
  off += (int64_t)n * npix * 16;
  Sel sel;
  sel.best = reinterpret_cast<unsigned long long*>(ws + off);
  off += (int64_t)n * 8;
  sel.count = reinterpret_cast<int*>(ws + off);
  hipLaunchKernelGGL(contour_sel_init_kernel, dim3((n + 63) / 64), dim3(64), 0, st, sel, n);
  const bool words = w % 4 == 0 && (reinterpret_cast<uintptr_t>(mask) & 3) == 0;  // rows then start on word boundaries
  if (words) {
    hipLaunchKernelGGL(cc_rows4_kernel, dim3((unsigned)((int64_t)n * h)), dim3(256), 0, st, mask, L, h, w, npix);
    hipLaunchKernelGGL(cc_union4_kernel, dim3(h, n), dim3(256), 0, st, mask, L, h, w, npix);
  } else {
    hipLaunchKernelGGL(cc_rows_kernel, dim3((unsigned)((int64_t)n * h)), dim3(256), 0, st, mask, L, h, w, npix);
    hipLaunchKernelGGL(cc_union_kernel, dim3(grid_for((int64_t)n * npix)), dim3(256), 0, st, mask, L, n, h, w, npix);
  }
  hipLaunchKernelGGL(cc_compress_kernel, dim3(grid_for((int64_t)n * (npix + 1))), dim3(256), 0, st, L, acc, n, npix);
  // (word-per-thread forms of this kernel measured slower, 222-292 vs 206 us per 16 masks: it is bound by the dependent label
  // lookups of the border pixels, which want one pixel per lane, not by its byte loads)
  hipLaunchKernelGGL(contour_sum_kernel, dim3(grid_for((int64_t)n * npix)), dim3(256), 0, st, mask, L, acc, n, h, w, npix);
  hipLaunchKernelGGL(contour_select_kernel, dim3(grid_for((int64_t)n * npix)), dim3(256), 0, st, mask, L, acc, sel, n, w, npix);
  hipLaunchKernelGGL(contour_final_kernel, dim3((n + 63) / 64), dim3(64), 0, st, acc, sel, reinterpret_cast<long long*>(out), n, npix);
  return lmx_launch_check("contour_final_kernel");
}
