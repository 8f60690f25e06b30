"""lmx_k_contour_features (csrc/contour.hip: connected components + crack sums on the device) against the sequential
border following of csrc/host_mask.cpp — the same 7 features of services/sam3-pipeline/app/main.py:102-145, BIT FOR BIT,
on analytic shapes, adversarial topologies and 1080p masks, in batches.  (cv2 is absent: both restate the published
algorithms; parity against real OpenCV stays unpinned.)"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KEYS = ["mask_area", "area_ratio", "circularity", "aspect_ratio", "centroid_x", "centroid_y", "perimeter"]


def _host(mask):
    from lmx import _lib

    m = np.ascontiguousarray(mask.astype(np.uint8))
    out = (C.c_double * 7)()
    assert _lib.load().lmx_h_mask_features(m.ctypes.data_as(C.c_void_p), m.shape[0], m.shape[1], C.cast(out, C.c_void_p)) == 0
    return dict(zip(KEYS, list(out)))


def _device(masks, cuda):
    """masks: bool/u8 [n,h,w] -> list of feature dicts, statistics from a plain count (mask_post is tested elsewhere)."""
    from lmx import kernels as K
    from lmx.services.sam3_pipeline import features_from_device

    m = np.ascontiguousarray(np.asarray(masks).astype(np.uint8))
    n, h, w = m.shape
    cont = K.contour_features(torch.from_numpy(m).to(cuda)).cpu().numpy()
    out = []
    for i in range(n):
        ys, xs = np.nonzero(m[i])
        stats = [len(xs), int(xs.sum()), int(ys.sum())]
        out.append(features_from_device(stats, cont[i], h, w))
    return out, cont


def _check(masks, cuda, label, oracle=False):
    """device == host C++ bit for bit; with oracle=True (small masks: the oracle is plain Python) also device == the
    independent restatement oracle/mask_features.py — integers (area, bounding box, step counts) exactly, the perimeter-derived
    doubles to 1e-12 (the oracle adds its steps in another order)."""
    from oracle import mask_features as OM

    dev, cont = _device(masks, cuda)
    for i, m in enumerate(masks):
        ref = _host(m)
        for k in KEYS:
            assert dev[i][k] == ref[k], f"{label}[{i}] {k}: device {dev[i][k]!r} host {ref[k]!r} (contour row {cont[i].tolist()})"
        if oracle:
            orc = OM.features(m)
            for k in KEYS:
                assert dev[i][k] == pytest.approx(orc[k], rel=1e-12, abs=1e-12), f"{label}[{i}] {k}: device {dev[i][k]!r} oracle {orc[k]!r}"
    return cont


def test_analytic_shapes_and_topologies(cuda):
    h, w = 60, 80
    shapes = []
    m = np.zeros((h, w), bool); m[10:30, 20:70] = True; shapes.append(m)                       # rectangle
    shapes.append(np.zeros((h, w), bool))                                                        # empty
    m = np.zeros((h, w), bool); m[4, 7] = True; shapes.append(m)                                 # single pixel
    m = np.zeros((h, w), bool); m[5, 2:9] = True; shapes.append(m)                               # one-pixel-wide line
    m = np.zeros((h, w), bool); m[2:8, 2:8] = True; m[15:45, 10:45] = True; m[20:40, 15:40] = False; m[25:35, 20:35] = True
    shapes.append(m)                                                                              # ring with an island in its hole
    shapes.append(np.ones((h, w), bool))                                                          # full frame
    m = np.zeros((h, w), bool); m[::2, ::2] = True; m[1::2, 1::2] = True; shapes.append(m)        # checkerboard: diagonal links only
    m = np.zeros((h, w), bool); m[np.arange(40), np.arange(40)] = True; shapes.append(m)          # a pure diagonal
    m = np.zeros((h, w), bool); m[0, :] = True; m[:, 0] = True; m[-1, :] = True; m[:, -1] = True; m[20:30, 20:30] = True
    shapes.append(m)                                                                              # frame ring: the inner square is NOT external
    yy, xx = np.mgrid[0:h, 0:w]
    shapes.append((xx - 40) ** 2 + (yy - 30) ** 2 <= 25 ** 2)                                     # disc
    m = np.zeros((h, w), bool); m[10:20, 10:20] = True; m[10:20, 30:40] = True; shapes.append(m)  # two equal squares: first in raster order wins
    m = np.zeros((h, w), bool); m[30:40, 10:20] = True; m[10:20, 30:41] = True; shapes.append(m)  # the larger one is later in raster order
    cont = _check(np.stack(shapes, 0), cuda, "shape", oracle=True)
    assert cont[0].tolist() == [2 * 49 * 19, 2 * (49 + 19), 0, 20, 10, 69, 29, 1]
    assert cont[1].tolist() == [0] * 8 and cont[2].tolist() == [0, 0, 0, 7, 4, 7, 4, 1]
    assert cont[4][7] == 2 and cont[8][7] == 1


@pytest.mark.parametrize("seed", range(8))
def test_random_blobs_with_salt_and_pepper(cuda, seed):
    rng = np.random.default_rng(seed)
    masks = []
    for _ in range(5):
        gh, gw = rng.integers(6, 20, 2)
        cell = int(rng.integers(2, 7))
        m = np.kron(rng.random((gh, gw)), np.ones((cell, cell))) > rng.uniform(0.4, 0.7)
        m ^= rng.random(m.shape) > 0.96  # thin structures, diagonal links, isolated pixels, pin holes
        masks.append(m)
    hh, ww = max(m.shape[0] for m in masks), max(m.shape[1] for m in masks)
    batch = np.zeros((len(masks), hh, ww), bool)
    for i, m in enumerate(masks):
        batch[i, :m.shape[0], :m.shape[1]] = m
    _check(batch, cuda, f"blobs seed {seed}", oracle=True)


def test_frozen_tie_break_and_arc_length_conventions(cuda):
    """The hand-derived cases of tests/test_mask_features.py (equal-area tie -> first component in raster order; all-isolated
    pixels; a pure diagonal) on the device kernel: the conventions cv2 cannot pin here are at least frozen in all three
    implementations."""
    import test_mask_features as TM

    for m, want in TM.frozen_tie_cases():
        dev, _ = _device(m[None], cuda)
        for k, v in want.items():
            assert dev[0][k] == pytest.approx(v, rel=1e-15, abs=0), (k, dev[0][k], v)


def test_1080p_masks_in_a_batch(cuda):
    """Masks at the service's size: smooth blobs (what SAM produces), one of them noisy, one touching the frame."""
    rng = np.random.default_rng(11)
    masks = []
    for j in range(4):
        base = rng.standard_normal((9, 16))
        big = np.kron(base, np.ones((120, 120)))
        # separable box blur by cumulative sums: smooth level sets
        k = 90
        c = np.cumsum(np.pad(big, ((k, k), (k, k)), mode="edge"), 0)
        big = (c[2 * k:] - c[:-2 * k])[:, k:-k]
        c = np.cumsum(np.pad(big, ((0, 0), (k, k)), mode="edge"), 1)
        big = c[:, 2 * k:] - c[:, :-2 * k]
        m = big > np.quantile(big, 0.6 + 0.1 * j)
        if j == 2:
            m ^= rng.random(m.shape) > 0.999
        masks.append(m[:1080, :1920])
    _check(np.stack(masks, 0), cuda, "1080p")


def test_service_features_from_device_records(cuda):
    """mask_post's statistics + the contour kernel give the service's feature dict without the mask leaving the GPU."""
    from lmx import kernels as K
    from lmx.services.sam3_pipeline import extract_segmentation_features, features_from_device

    g = torch.Generator(device=cuda).manual_seed(3)
    low = torch.nn.functional.interpolate(torch.randn(3, 1, 12, 12, device=cuda, generator=g), size=(256, 256), mode="bilinear")[:, 0]
    mask, stats = K.mask_post(low.contiguous(), 1024, 576, 1024, 1080, 1920)
    cont = K.contour_features(mask)
    torch.cuda.synchronize()
    mh, sh, ch = mask.cpu().numpy(), stats.cpu().numpy(), cont.cpu().numpy()
    for i in range(3):
        assert features_from_device(sh[i], ch[i], 1080, 1920) == extract_segmentation_features(mh[i])
