// host_track.cpp — HOST-side association arithmetic of the tracking service, the consumer directly after the hot path
// (SURVEY.md section 8f rank 4): services/tracking-service/app/tracker/matching.py:12-44 (iou_batch) and :69-101
// (linear_assignment = lap.lapjv(cost, extend_cost=True, cost_limit=100000)).  `lap` is a third-party package that is
// neither in /root/reference nor installed; what it computes there is the minimum-cost assignment of the rectangular
// matrix that matches min(rows, cols) pairs (the cost limit of 1e5 never binds on costs in [0, 2]), restated here as
// the shortest-augmenting-path Hungarian method with dual potentials.  Sizes are detections x tracks of one frame
// (tens), the work is sequential per frame: host code, like the reference's.  PARITY UNPINNED against lap itself
// (tests check optimality against scipy.optimize.linear_sum_assignment).
#include <math.h>
#include <stdint.h>

#include <limits>
#include <vector>

#include "../../include/lmx.h"

extern "C" int lmx_h_iou_matrix(const double* a, int n, const double* b, int m, double* out) {
  if ((n > 0 && !a) || (m > 0 && !b) || (n > 0 && m > 0 && !out) || n < 0 || m < 0) return LMX_EINVAL;
  for (int i = 0; i < n; ++i) {
    const double* p = a + 4 * i;
    const double area_a = (p[2] - p[0]) * (p[3] - p[1]);
    for (int j = 0; j < m; ++j) {
      const double* q = b + 4 * j;
      const double xx1 = p[0] > q[0] ? p[0] : q[0], yy1 = p[1] > q[1] ? p[1] : q[1];
      const double xx2 = p[2] < q[2] ? p[2] : q[2], yy2 = p[3] < q[3] ? p[3] : q[3];
      const double w = xx2 - xx1 > 0.0 ? xx2 - xx1 : 0.0, h = yy2 - yy1 > 0.0 ? yy2 - yy1 : 0.0;
      const double inter = w * h;
      const double area_b = (q[2] - q[0]) * (q[3] - q[1]);
      out[(int64_t)i * m + j] = inter / ((area_a + area_b - inter) + 1e-6);  // matching.py:40-42
    }
  }
  return LMX_OK;
}

namespace {

// rows <= cols; cost(i, j) given by a functor; row_to_col[rows] receives the assignment
template <class Cost>
void hungarian(int rows, int cols, Cost cost, int* row_to_col) {
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<double> u(rows + 1, 0.0), v(cols + 1, 0.0), minv(cols + 1);
  std::vector<int> p(cols + 1, 0), way(cols + 1, 0);
  std::vector<char> used(cols + 1);
  for (int i = 1; i <= rows; ++i) {
    p[0] = i;
    int j0 = 0;
    std::fill(minv.begin(), minv.end(), INF);
    std::fill(used.begin(), used.end(), 0);
    do {
      used[j0] = 1;
      const int i0 = p[j0];
      double delta = INF;
      int j1 = 0;
      for (int j = 1; j <= cols; ++j) {
        if (used[j]) continue;
        const double cur = cost(i0 - 1, j - 1) - u[i0] - v[j];
        if (cur < minv[j]) {
          minv[j] = cur;
          way[j] = j0;
        }
        if (minv[j] < delta) {
          delta = minv[j];
          j1 = j;
        }
      }
      for (int j = 0; j <= cols; ++j) {
        if (used[j]) {
          u[p[j]] += delta;
          v[j] -= delta;
        } else {
          minv[j] -= delta;
        }
      }
      j0 = j1;
    } while (p[j0] != 0);
    do {
      const int j1 = way[j0];
      p[j0] = p[j1];
      j0 = j1;
    } while (j0);
  }
  for (int j = 1; j <= cols; ++j)
    if (p[j]) row_to_col[p[j] - 1] = j - 1;
}

}  // namespace

extern "C" int lmx_h_assign(const double* cost, int n, int m, int* row_to_col, int* col_to_row) {
  if (n < 0 || m < 0 || (n > 0 && !row_to_col) || (m > 0 && !col_to_row) || (n > 0 && m > 0 && !cost)) return LMX_EINVAL;
  for (int i = 0; i < n; ++i) row_to_col[i] = -1;
  for (int j = 0; j < m; ++j) col_to_row[j] = -1;
  if (n == 0 || m == 0) return LMX_OK;
  for (int64_t k = 0; k < (int64_t)n * m; ++k)
    if (!isfinite(cost[k])) return LMX_EINVAL;
  if (n <= m) {
    hungarian(n, m, [&](int i, int j) { return cost[(int64_t)i * m + j]; }, row_to_col);
    for (int i = 0; i < n; ++i) col_to_row[row_to_col[i]] = i;
  } else {
    hungarian(m, n, [&](int j, int i) { return cost[(int64_t)i * m + j]; }, col_to_row);
    for (int j = 0; j < m; ++j) row_to_col[col_to_row[j]] = j;
  }
  return LMX_OK;
}
