"""Host mask features (lmx_h_mask_features, C++) vs the independent Python restatement and closed-form answers.
cv2 is not installed: these pin the restated semantics (Suzuki-Abe outer borders, Green area, chain perimeter), not
real OpenCV output (parity unpinned)."""
import ctypes as C
import math

import numpy as np
import pytest

from lmx import _lib
from oracle import mask_features as OM

KEYS = ["mask_area", "area_ratio", "circularity", "aspect_ratio", "centroid_x", "centroid_y", "perimeter"]


def _c(mask):
    m = np.ascontiguousarray(mask.astype(np.uint8))
    out = (C.c_double * 7)()
    rc = _lib.load().lmx_h_mask_features(m.ctypes.data_as(C.c_void_p), m.shape[0], m.shape[1], C.cast(out, C.c_void_p))
    assert rc == 0
    return dict(zip(KEYS, list(out)))


def test_rectangle_closed_form():
    m = np.zeros((60, 80), bool)
    m[10:30, 20:70] = True  # 20 rows x 50 cols
    f = _c(m)
    assert f["mask_area"] == 1000 and f["perimeter"] == 2 * (49 + 19)
    assert f["aspect_ratio"] == 50 / 20 and f["centroid_x"] == 44.5 and f["centroid_y"] == 19.5
    assert math.isclose(f["circularity"], 4 * math.pi * (49 * 19) / (2 * (49 + 19)) ** 2)


def test_empty_single_pixel_and_line():
    f = _c(np.zeros((10, 12), bool))
    assert f["mask_area"] == 0 and f["perimeter"] == 0 and f["centroid_x"] == 6 and f["centroid_y"] == 5
    m = np.zeros((10, 12), bool)
    m[4, 7] = True
    f = _c(m)
    assert f["mask_area"] == 1 and f["perimeter"] == 0 and f["circularity"] == 0 and f["aspect_ratio"] == 1
    m = np.zeros((10, 12), bool)
    m[5, 2:9] = True  # a 7-pixel horizontal line: the border goes out and back
    f = _c(m)
    assert f["perimeter"] == 12 and f["circularity"] == 0 and f["aspect_ratio"] == 7


def test_largest_of_several_and_nested_component_is_not_external():
    m = np.zeros((50, 50), bool)
    m[2:8, 2:8] = True          # small square
    m[15:45, 10:45] = True      # big ring ...
    m[20:40, 15:40] = False     # ... with a hole
    m[25:35, 20:35] = True      # island inside the hole: NOT an external contour
    f = _c(m)
    assert f["perimeter"] == 2 * (34 + 29) and f["aspect_ratio"] == 35 / 30
    assert f == pytest.approx(OM.features(m))


@pytest.mark.parametrize("seed", range(6))
def test_random_blobs_match_python_restatement(seed):
    rng = np.random.default_rng(seed)
    base = rng.random((12, 16))
    m = np.kron(base, np.ones((4, 4))) > 0.55
    m ^= rng.random(m.shape) > 0.97  # salt and pepper: thin structures, diagonal links, isolated pixels
    a, b = _c(m), OM.features(m)
    for k in KEYS:
        assert a[k] == pytest.approx(b[k], rel=1e-12, abs=1e-12), k


def test_disc_is_nearly_circular():
    yy, xx = np.mgrid[0:200, 0:200]
    m = (xx - 100) ** 2 + (yy - 90) ** 2 <= 60 ** 2
    f = _c(m)
    assert 0.85 < f["circularity"] < 1.0 and abs(f["centroid_x"] - 100) < 1e-9 and abs(f["aspect_ratio"] - 1) < 1e-9


# ---- frozen conventions where cv2's behaviour is NOT pinned (cv2 absent; DESIGN.md section 4 "parity unpinned") --------------------
# Hand-derived expectations, so that the chosen rules cannot drift between the host code, the device kernel and the oracle:
#  * equal contourArea: max(contours, key=cv2.contourArea) returns the first maximal element of the list findContours returned;
#    this restatement takes the component whose first pixel comes FIRST IN RASTER ORDER (OpenCV's list order is an
#    implementation detail — historically the reverse scan order — so a real cv2 may pick the other one on an exact tie);
#  * perimeter = (unit steps) + (diagonal steps) * sqrt(2) in double (cv2.arcLength sums float32 segment lengths of the
#    CHAIN_APPROX_SIMPLE polygon: equal for axis-aligned contours, ~1e-8 relative apart on diagonal runs).
def frozen_tie_cases():
    a = np.zeros((40, 40), bool)
    a[2:12, 3:8] = True      # 10 rows x 5 cols, first in raster order: contourArea = 4 * 9 = 36
    a[20:25, 10:20] = True   # 5 rows x 10 cols: contourArea = 9 * 4 = 36  (an exact tie)
    b = np.zeros((12, 12), bool)
    b[3, 5] = True           # isolated pixels only: every contourArea is 0
    b[7, 2] = True
    b[9, 9] = True
    c = np.zeros((12, 12), bool)
    c[np.arange(2, 8), np.arange(3, 9)] = True  # a pure diagonal of 6 pixels: out and back = 10 diagonal steps
    return [
        (a, dict(mask_area=100.0, area_ratio=100 / 1600, perimeter=2.0 * (4 + 9), aspect_ratio=5 / 10,
                 circularity=4 * math.pi * 36 / (2.0 * (4 + 9)) ** 2, centroid_x=(50 * 5 + 50 * 14.5) / 100, centroid_y=(50 * 6.5 + 50 * 22) / 100)),
        (b, dict(mask_area=3.0, area_ratio=3 / 144, perimeter=0.0, aspect_ratio=1.0, circularity=0.0, centroid_x=16 / 3, centroid_y=19 / 3)),
        (c, dict(mask_area=6.0, area_ratio=6 / 144, perimeter=10 * math.sqrt(2.0), aspect_ratio=1.0, circularity=0.0, centroid_x=5.5, centroid_y=4.5)),
    ]


def test_frozen_tie_break_and_arc_length_conventions():
    for m, want in frozen_tie_cases():
        for impl in (_c(m), OM.features(m)):
            for k, v in want.items():
                assert impl[k] == pytest.approx(v, rel=1e-15, abs=0), (k, impl[k], v)
