"""Host-side geometry and tables of the YOLO preprocessing (ultralytics LetterBox + cv2.resize INTER_LINEAR), and the
inverse map (scale_boxes) — everything the predictor does around the network under
services/yolo-pipeline/app/main.py:76.  ultralytics and cv2 are not installed and not in /root/reference: these
restate their published algorithms (ultralytics/data/augment.py LetterBox.__call__, ultralytics/utils/ops.py
scale_boxes, OpenCV modules/imgproc/src/resize.cpp resizeGeneric_ / HResizeLinear / VResizeLinear for 8U).
PARITY UNPINNED against real cv2; the device kernel is checked against the numpy restatement below.
"""
from dataclasses import dataclass

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


@dataclass(frozen=True)
class LetterboxGeo:
    sh: int
    sw: int
    rh: int       # resized (unpadded) size
    rw: int
    top: int
    left: int
    oh: int       # network input size
    ow: int
    gain: float   # min(oh/sh, ow/sw) as scale_boxes recomputes it
    pad_x: float
    pad_y: float


def geometry(sh, sw, imgsz=640, stride=32, auto=True):
    """LetterBox(new_shape=imgsz, auto, scaleFill=False, scaleup=True, center=True, stride)."""
    new_h = new_w = int(imgsz)
    r = min(new_h / sh, new_w / sw)
    rw, rh = int(round(sw * r)), int(round(sh * r))
    dw, dh = new_w - rw, new_h - rh
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    oh, ow = rh + top + bottom, rw + left + right
    # ops.scale_boxes(img1_shape=(oh,ow), boxes, img0_shape=(sh,sw)): gain and pad are re-derived from the shapes
    gain = min(oh / sh, ow / sw)
    pad_x = round((ow - sw * gain) / 2 - 0.1)
    pad_y = round((oh - sh * gain) / 2 - 0.1)
    return LetterboxGeo(sh, sw, rh, rw, top, left, oh, ow, gain, float(pad_x), float(pad_y))


def _axis_table(ssize, dsize):
    """OpenCV resizeGeneric_ table build for INTER_LINEAR, 8U fixed point: (ofs int32 [d], coef int16 [d][2])."""
    inv_scale = float(dsize) / float(ssize)
    scale = 1.0 / inv_scale
    ofs = np.zeros(dsize, np.int32)
    coef = np.zeros((dsize, 2), np.int16)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        ofs[d] = s
        c0 = np.float32(1.0) - f
        # saturate_cast<short>(float) = cvRound (round half to even) then saturate
        coef[d, 0] = np.clip(np.rint(np.float32(c0 * np.float32(INTER_RESIZE_COEF_SCALE))), -32768, 32767)
        coef[d, 1] = np.clip(np.rint(np.float32(f * np.float32(INTER_RESIZE_COEF_SCALE))), -32768, 32767)
    return ofs, coef


def resize_tables(sh, sw, rh, rw):
    """-> (xofs, ialpha, yofs, ibeta).  x: the table build clamps (sx<0 -> sx=0,fx=0; sx>=sw-1 -> sx=sw-1,fx=0);
    y: offsets are kept raw and the row index is clipped when the rows are fetched."""
    xofs, ialpha = _axis_table(sw, rw)
    for d in range(rw):
        if xofs[d] < 0:
            xofs[d] = 0
            ialpha[d] = (INTER_RESIZE_COEF_SCALE, 0)
        if xofs[d] >= sw - 1:
            xofs[d] = sw - 1
            ialpha[d] = (INTER_RESIZE_COEF_SCALE, 0)
    yofs, ibeta = _axis_table(sh, rh)
    return xofs, ialpha.reshape(-1).copy(), yofs, ibeta.reshape(-1).copy()


def letterbox_reference(frame, geo, swap_rb=True):
    """numpy restatement of the device kernel (used by tests; also the oracle's LetterBox)."""
    sh, sw = frame.shape[:2]
    if (geo.rh, geo.rw) != (sh, sw):
        xofs, ialpha, yofs, ibeta = resize_tables(sh, sw, geo.rh, geo.rw)
        ia = ialpha.reshape(-1, 2).astype(np.int32)
        ib = ibeta.reshape(-1, 2).astype(np.int32)
        x0 = xofs
        x1 = np.minimum(xofs + 1, sw - 1)
        src = frame.astype(np.int32)
        hbuf = src[:, x0, :] * ia[None, :, 0, None] + src[:, x1, :] * ia[None, :, 1, None]  # [sh, rw, 3]
        y0 = np.clip(yofs, 0, sh - 1)
        y1 = np.clip(yofs + 1, 0, sh - 1)
        val = (((ib[:, 0, None, None] * (hbuf[y0] >> 4)) >> 16) + ((ib[:, 1, None, None] * (hbuf[y1] >> 4)) >> 16) + 2) >> 2
        img = np.clip(val, 0, 255).astype(np.uint8)
    else:
        img = frame
    out = np.full((geo.oh, geo.ow, 3), 114, np.uint8)
    out[geo.top:geo.top + geo.rh, geo.left:geo.left + geo.rw] = img
    return out[:, :, ::-1].copy() if swap_rb else out
