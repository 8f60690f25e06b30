"""Host-side tables for the preprocessing kernels.

``coeff_tables`` restates Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (Pillow src/libImaging/Resample.c,
not in this tree; checked bit-for-bit against the installed Pillow in tests/test_resample.py) so the device kernels
(lmx_k_pil_resize_h/_v) reproduce ``PIL.Image.resize`` on uint8 exactly.  Used for
  * DINO: shortest-edge-256 BICUBIC (AutoImageProcessor, services/dinov3-pipeline/app/main.py:107)
  * SAM : ResizeLongestSide(1024) BILINEAR via torchvision ``resize(to_pil_image(..))`` (services/sam3-pipeline/app/main.py:80)
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = "bilinear", "bicubic"


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


_FILTERS = {BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0)}


def coeff_tables(in_size, out_size, filt):
    """-> (bounds int32 [out*2] = (xmin, count), kk int32 [out*ksize], ksize) for one axis (box = full input)."""
    fn, fsupport = _FILTERS[filt]
    in0, in1 = 0.0, float(in_size)
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ww = 0.0
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        for x in range(xmax):
            w = fn((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    # normalize_coeffs_8bpc: (int)(+-0.5 + k * 2^22), C truncation toward zero
    scaled = kk * float(1 << PRECISION_BITS)
    ikk = np.where(kk < 0, np.trunc(-0.5 + scaled), np.trunc(0.5 + scaled)).astype(np.int32)
    return bounds.reshape(-1), ikk.reshape(-1), ksize


def resize_u8_reference(img, dw, dh, filt):
    """numpy restatement of the two device passes (used by the CPU tests to pin the tables against PIL)."""
    h, w, _ = img.shape
    cur = img
    if dw != w:
        b, k, ks = coeff_tables(w, dw, filt)
        b = b.reshape(-1, 2)
        k = k.reshape(-1, ks)
        out = np.empty((h, dw, 3), np.uint8)
        for xo in range(dw):
            xmin, cnt = b[xo]
            acc = (cur[:, xmin:xmin + cnt, :].astype(np.int64) * k[xo, :cnt, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xo, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = out
    if dh != h:
        b, k, ks = coeff_tables(h, dh, filt)
        b = b.reshape(-1, 2)
        k = k.reshape(-1, ks)
        out = np.empty((dh, cur.shape[1], 3), np.uint8)
        for yo in range(dh):
            ymin, cnt = b[yo]
            acc = (cur[ymin:ymin + cnt].astype(np.int64) * k[yo, :cnt, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yo] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = out
    return cur


def shortest_edge_size(h, w, edge):
    """transformers get_resize_output_image_size(default_to_square=False): (new_h, new_w), int() truncation."""
    short, long = (w, h) if w <= h else (h, w)
    if short == edge:
        return h, w
    new_short, new_long = edge, int(edge * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def norm_lut(mean, std, rescale=1.0 / 255.0):
    """lut[c][u]: the image processor's rescale + normalize applied to byte u (same numpy expressions as
    transformers.image_transforms.rescale/normalize: f64 multiply -> f32, then (x - mean)/std in f32)."""
    u = np.arange(256, dtype=np.uint8)
    x = (u.astype(np.float64) * rescale).astype(np.float32)
    m = np.array(mean, dtype=np.float32)
    s = np.array(std, dtype=np.float32)
    return ((x[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
