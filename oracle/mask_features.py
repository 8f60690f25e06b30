"""oracle.mask_features — plain-Python restatement of extract_segmentation_features
(services/sam3-pipeline/app/main.py:102-145) for SMALL masks: the same published algorithms as csrc/host_mask.cpp
(Suzuki-Abe outer border following, Green's formula, chain length), written independently with explicit point lists so
the C++ host code can be cross-checked, plus closed-form answers for analytic shapes (tests/test_mask_features.py).
cv2 is not installed: PARITY UNPINNED against real OpenCV.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import math

import numpy as np

_NB = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]  # clockwise from west, (dx, dy)


def _outer_border(m, sx, sy):
    h, w = m.shape

    def fg(x, y):
        return 0 <= x < w and 0 <= y < h and m[y, x]

    first = None
    for dx, dy in _NB:
        if fg(sx + dx, sy + dy):
            first = (sx + dx, sy + dy)
            break
    if first is None:
        return [(sx, sy)]
    pts = [(sx, sy)]
    prev, cur = first, (sx, sy)
    while True:
        k0 = _NB.index((prev[0] - cur[0], prev[1] - cur[1]))
        nxt = cur
        for s in range(1, 9):
            dx, dy = _NB[(k0 - s) % 8]
            if fg(cur[0] + dx, cur[1] + dy):
                nxt = (cur[0] + dx, cur[1] + dy)
                break
        if nxt == (sx, sy) and cur == first:
            break
        pts.append(nxt)
        prev, cur = cur, nxt
    return pts  # closed chain: the edge pts[-1] -> pts[0] is implied


def _components(m):
    """8-connected components in raster order of their first pixel; yields (start_x, start_y, set_of_pixels)."""
    h, w = m.shape
    seen = np.zeros_like(m, bool)
    for y in range(h):
        for x in range(w):
            if m[y, x] and not seen[y, x]:
                comp, st = [], [(x, y)]
                seen[y, x] = True
                while st:
                    cx, cy = st.pop()
                    comp.append((cx, cy))
                    for dx, dy in _NB:
                        qx, qy = cx + dx, cy + dy
                        if 0 <= qx < w and 0 <= qy < h and m[qy, qx] and not seen[qy, qx]:
                            seen[qy, qx] = True
                            st.append((qx, qy))
                yield x, y, comp


def _outside_background(m):
    h, w = m.shape
    out = np.zeros_like(m, bool)
    st = [(x, y) for x in range(w) for y in (0, h - 1)] + [(x, y) for y in range(h) for x in (0, w - 1)]
    st = [(x, y) for x, y in st if not m[y, x]]
    for x, y in st:
        out[y, x] = True
    while st:
        x, y = st.pop()
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            qx, qy = x + dx, y + dy
            if 0 <= qx < w and 0 <= qy < h and not m[qy, qx] and not out[qy, qx]:
                out[qy, qx] = True
                st.append((qx, qy))
    return out


def features(mask):
    m = np.asarray(mask).astype(bool)
    h, w = m.shape
    area = float(m.sum())
    outside = _outside_background(m)
    best = None
    for sx, sy, _ in _components(m):
        if not (sx == 0 or outside[sy, sx - 1]):
            continue  # nested inside a hole: not an EXTERNAL contour
        pts = _outer_border(m, sx, sy)
        n = len(pts)
        a2 = sum(pts[i][0] * pts[(i + 1) % n][1] - pts[(i + 1) % n][0] * pts[i][1] for i in range(n))
        # chain length = unit steps + diagonal steps * sqrt 2 (counted, so that the sum does not depend on the order)
        steps = [(abs(pts[(i + 1) % n][0] - pts[i][0]), abs(pts[(i + 1) % n][1] - pts[i][1])) for i in range(n)] if n > 1 else []
        per = float(sum(1 for s_ in steps if s_[0] + s_[1] == 1)) + float(sum(1 for s_ in steps if s_[0] + s_[1] == 2)) * math.sqrt(2.0)
        xs, ys = [p[0] for p in pts], [p[1] for p in pts]
        cand = (abs(a2) / 2.0, per, max(xs) - min(xs) + 1, max(ys) - min(ys) + 1)
        if best is None or cand[0] > best[0]:
            best = cand
    ys, xs = np.nonzero(m)
    out = {"mask_area": area, "area_ratio": area / (h * w) if h * w else 0.0}
    if best:
        a, per, bw, bh = best
        out["circularity"] = (4 * math.pi * a) / (per ** 2) if per > 0 else 0.0
        out["aspect_ratio"] = bw / bh if bh > 0 else 0.0
        out["perimeter"] = per
    else:
        out["circularity"], out["aspect_ratio"], out["perimeter"] = 0.0, 0.0, 0.0
    out["centroid_x"] = float(xs.sum()) / area if area else w / 2
    out["centroid_y"] = float(ys.sum()) / area if area else h / 2
    return out
