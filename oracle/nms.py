"""oracle.nms — CPU restatement of ultralytics.utils.ops.non_max_suppression as the predictor calls it under
services/yolo-pipeline/app/main.py:76 (conf from config, iou 0.7, max_det 300, single label, class offset 7680) and of
torchvision.ops.nms (csrc/ops/cpu/nms_kernel.cpp).  Neither package is installed and the reference holds no NMS
vectors: PARITY UNPINNED beyond the authored known-answer cases in tests/test_nms_oracle.py.
TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import numpy as np


def torchvision_nms(boxes, scores, iou_threshold):
    """boxes f32 [n,4] xyxy, scores f32 [n] -> kept indices in decreasing score order (greedy, suppress IoU > thr).
    Arithmetic in float32 step by step, threshold compared in double, as the C++ CPU kernel does."""
    boxes = np.asarray(boxes, np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    x1, y1, x2, y2 = (boxes[:, i] for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    suppressed = np.zeros(n, bool)
    keep = []
    thr = float(iou_threshold)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        rest = rest[~suppressed[rest]]
        if rest.size == 0:
            continue
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return np.asarray(keep, np.int64)


def non_max_suppression(pred, conf, iou=0.7, max_det=300, max_wh=7680.0, max_nms=30000):
    """pred f32 [A, 4+nc] (cx,cy,w,h, class scores) for ONE image -> (boxes [k,4] xyxy, scores [k], cls [k], src [k])."""
    pred = np.asarray(pred, np.float32)
    cls_scores = pred[:, 4:]
    best = cls_scores.max(1)
    cand = np.nonzero(best > np.float32(conf))[0]
    if cand.size == 0:
        z = np.zeros((0,), np.float32)
        return np.zeros((0, 4), np.float32), z, np.zeros((0,), np.int64), np.zeros((0,), np.int64)
    x = pred[cand]
    dw, dh = x[:, 2] / np.float32(2), x[:, 3] / np.float32(2)
    box = np.stack([x[:, 0] - dw, x[:, 1] - dh, x[:, 0] + dw, x[:, 1] + dh], 1).astype(np.float32)
    j = cls_scores[cand].argmax(1)  # first maximum, like torch.max
    sc = best[cand]
    order = np.argsort(-sc, kind="stable")[:max_nms]  # ties: lower anchor index first (documented choice)
    box, sc, j, cand = box[order], sc[order], j[order], cand[order]
    off = (j.astype(np.float32) * np.float32(max_wh))[:, None]
    keep = torchvision_nms(box + off, sc, iou)[:max_det]
    return box[keep], sc[keep], j[keep].astype(np.int64), cand[keep].astype(np.int64)
