// host_mask.cpp — HOST-side mask features of services/sam3-pipeline/app/main.py:102-145 (extract_segmentation_features):
// np.sum / cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) / max by cv2.contourArea / cv2.arcLength /
// cv2.boundingRect / cv2.moments.  cv2 is not in /root/reference nor installed; this restates the published algorithms
// (Suzuki-Abe border following for the outer border of each 8-connected component, Green's formula for contourArea,
// Euclidean chain length for arcLength).  Border following is sequential and irregular (SURVEY.md §7 "hard parts"): it
// runs on the host over the 1-byte-per-pixel mask the GPU wrote, ~1-3 ms per 1080p mask.  PARITY UNPINNED vs real cv2.
//
// CHAIN_APPROX_SIMPLE only drops collinear interior points of straight runs, so polygon area, perimeter and bounding box
// equal those of the full border chain; the chain itself is what is traced here.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/lmx.h"

namespace {

struct ContourStat {
  double area = 0.0, perimeter = 0.0;
  int minx = 0, miny = 0, maxx = 0, maxy = 0;
};

// 8-neighbourhood in clockwise order starting from west (x-1,y): W, NW, N, NE, E, SE, S, SW
const int DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

inline bool on(const uint8_t* m, int h, int w, int x, int y) { return x >= 0 && y >= 0 && x < w && y < h && m[(int64_t)y * w + x] != 0; }

// Suzuki-Abe outer border following from start pixel (sx,sy) whose west neighbour is background.
ContourStat trace_outer(const uint8_t* m, int h, int w, int sx, int sy) {
  ContourStat st;
  st.minx = st.maxx = sx;
  st.miny = st.maxy = sy;
  // (3.1) clockwise search around the start, beginning at the west neighbour, for the first foreground pixel
  int d1 = -1;
  for (int k = 0; k < 8; ++k) {
    if (on(m, h, w, sx + DX[k], sy + DY[k])) {
      d1 = k;
      break;
    }
  }
  if (d1 < 0) return st;  // isolated pixel: area 0, perimeter 0, 1x1 box
  const int x1 = sx + DX[d1], y1 = sy + DY[d1];
  int px = x1, py = y1;  // (i2,j2): previous border pixel
  int cx = sx, cy = sy;  // (i3,j3): current border pixel
  double twice_area = 0.0;
  int64_t n_unit = 0, n_diag = 0;  // chain steps by length: the perimeter is n_unit + n_diag * sqrt 2 (order-free, like the device form)
  for (int64_t guard = 0; guard < (int64_t)8 * h * w + 16; ++guard) {
    // (3.3) counter-clockwise search around the current pixel, starting after the direction of the previous pixel
    int dprev = 0;
    for (int k = 0; k < 8; ++k)
      if (cx + DX[k] == px && cy + DY[k] == py) dprev = k;
    int nx = cx, ny = cy;
    for (int s = 1; s <= 8; ++s) {
      const int k = (dprev - s + 16) & 7;  // counter-clockwise = decreasing index in the clockwise table
      if (on(m, h, w, cx + DX[k], cy + DY[k])) {
        nx = cx + DX[k];
        ny = cy + DY[k];
        break;
      }
    }
    // edge current -> next contributes to the shoelace sum and the chain length
    twice_area += (double)cx * ny - (double)nx * cy;
    if (nx != cx && ny != cy)
      ++n_diag;
    else if (nx != cx || ny != cy)
      ++n_unit;
    if (nx < st.minx) st.minx = nx;
    if (nx > st.maxx) st.maxx = nx;
    if (ny < st.miny) st.miny = ny;
    if (ny > st.maxy) st.maxy = ny;
    // (3.5) stop when we are back at the start AND the next step repeats the first one
    if (nx == sx && ny == sy && cx == x1 && cy == y1) break;
    px = cx;
    py = cy;
    cx = nx;
    cy = ny;
  }
  st.area = fabs(twice_area) * 0.5;
  st.perimeter = (double)n_unit + (double)n_diag * 1.41421356237309504880;
  return st;
}

}  // namespace

// out[7] = mask_area, area_ratio, circularity, aspect_ratio, centroid_x, centroid_y, perimeter (the dict of sam3 main.py:137-145)
extern "C" int lmx_h_mask_features(const uint8_t* mask, int h, int w, double* out) {
  if (!mask || !out || h <= 0 || w <= 0) return LMX_EINVAL;
  const int64_t N = (int64_t)h * w;
  // moments of the 0/1 image (cv2.moments(mask.astype(uint8))) and the pixel count (np.sum)
  double m00 = 0, m10 = 0, m01 = 0;
  for (int y = 0; y < h; ++y) {
    const uint8_t* r = mask + (int64_t)y * w;
    int64_t c = 0, sx = 0;
    for (int x = 0; x < w; ++x)
      if (r[x]) {
        ++c;
        sx += x;
      }
    m00 += (double)c;
    m10 += (double)sx;
    m01 += (double)c * y;
  }
  // "outside" background = background 4-connected to the image frame; RETR_EXTERNAL keeps only components bordering it
  std::vector<uint8_t> lab((size_t)N, 0);  // 1 = outside background, 2 = visited foreground
  std::vector<int> stack;
  auto push_bg = [&](int x, int y) {
    const int64_t i = (int64_t)y * w + x;
    if (!mask[i] && !lab[i]) {
      lab[i] = 1;
      stack.push_back((int)i);
    }
  };
  for (int x = 0; x < w; ++x) {
    push_bg(x, 0);
    push_bg(x, h - 1);
  }
  for (int y = 0; y < h; ++y) {
    push_bg(0, y);
    push_bg(w - 1, y);
  }
  while (!stack.empty()) {
    const int i = stack.back();
    stack.pop_back();
    const int x = i % w, y = i / w;
    if (x > 0) push_bg(x - 1, y);
    if (x + 1 < w) push_bg(x + 1, y);
    if (y > 0) push_bg(x, y - 1);
    if (y + 1 < h) push_bg(x, y + 1);
  }
  bool any = false;
  ContourStat best;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const int64_t i = (int64_t)y * w + x;
      if (!mask[i] || lab[i]) continue;
      // first pixel of a new 8-connected component in raster order: its west neighbour is background or the frame
      const bool external = (x == 0) || lab[i - 1] == 1;
      // flood the component (8-connectivity) so that it is visited once
      lab[i] = 2;
      stack.push_back((int)i);
      while (!stack.empty()) {
        const int j = stack.back();
        stack.pop_back();
        const int jx = j % w, jy = j / w;
        for (int k = 0; k < 8; ++k) {
          const int qx = jx + DX[k], qy = jy + DY[k];
          if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
          const int64_t q = (int64_t)qy * w + qx;
          if (mask[q] && !lab[q]) {
            lab[q] = 2;
            stack.push_back((int)q);
          }
        }
      }
      if (!external) continue;
      const ContourStat st = trace_outer(mask, h, w, x, y);
      if (!any || st.area > best.area) best = st;  // max(contours, key=contourArea): first maximum in raster order
      any = true;
    }
  }
  const double PI = 3.14159265358979323846;
  out[0] = m00;
  out[1] = N > 0 ? m00 / (double)N : 0.0;
  if (any) {
    out[2] = best.perimeter > 0 ? (4.0 * PI * best.area) / (best.perimeter * best.perimeter) : 0.0;
    const int bw = best.maxx - best.minx + 1, bh = best.maxy - best.miny + 1;
    out[3] = bh > 0 ? (double)bw / (double)bh : 0.0;
    out[6] = best.perimeter;
  } else {
    out[2] = 0.0;
    out[3] = 0.0;
    out[6] = 0.0;
  }
  if (m00 != 0) {
    out[4] = m10 / m00;
    out[5] = m01 / m00;
  } else {
    out[4] = w / 2.0;
    out[5] = h / 2.0;
  }
  return LMX_OK;
}
