"""Authored known-answer cases for oracle.nms (no importable ultralytics/torchvision: SURVEY.md §8c — these pin the
semantics the restatement claims, they are not reference-generated vectors)."""
import numpy as np

from oracle import nms as ON


def _pred(boxes_xyxy, scores, classes, nc=4):
    b = np.asarray(boxes_xyxy, np.float32)
    p = np.zeros((len(b), 4 + nc), np.float32)
    p[:, 0] = (b[:, 0] + b[:, 2]) / 2
    p[:, 1] = (b[:, 1] + b[:, 3]) / 2
    p[:, 2] = b[:, 2] - b[:, 0]
    p[:, 3] = b[:, 3] - b[:, 1]
    for i, (s, c) in enumerate(zip(scores, classes)):
        p[i, 4 + c] = s
    return p


def test_greedy_suppression_and_order():
    # box1 overlaps box0 with IoU 0.81 (suppressed), box2 is disjoint
    p = _pred([[0, 0, 100, 100], [0, 0, 100, 81], [200, 200, 300, 300]], [0.9, 0.8, 0.7], [0, 0, 0])
    box, sc, cls, src = ON.non_max_suppression(p, 0.25)
    assert src.tolist() == [0, 2] and np.allclose(sc, [0.9, 0.7])


def test_iou_exactly_at_threshold_is_kept():
    # IoU = 70/100 = 0.7 exactly in f32 -> (double)0.7f > 0.7 is True for f32 0.7 = 0.699999988? no: f32(0.7) < 0.7
    p = _pred([[0, 0, 100, 100], [0, 0, 100, 70]], [0.9, 0.8], [0, 0])
    _, _, _, src = ON.non_max_suppression(p, 0.25)
    ovr = np.float32(7000.0) / np.float32(10000.0)
    assert (float(ovr) > 0.7) is False  # f32(0.7) = 0.699999988 < 0.7 (double)  -> not suppressed
    assert src.tolist() == [0, 1]


def test_class_offset_separates_classes():
    p = _pred([[0, 0, 100, 100], [0, 0, 100, 100]], [0.9, 0.8], [0, 1])
    _, _, cls, src = ON.non_max_suppression(p, 0.25)
    assert src.tolist() == [0, 1] and cls.tolist() == [0, 1]


def test_conf_filter_is_strict_and_uses_best_class():
    p = _pred([[0, 0, 10, 10], [20, 20, 30, 30], [40, 40, 50, 50]], [0.5, 0.5000001, 0.2], [1, 2, 3])
    p[2, 4 + 0] = 0.6  # a second class pushes box 2 over the threshold with class 0
    _, sc, cls, src = ON.non_max_suppression(p, 0.5)
    assert src.tolist() == [2, 1] and cls.tolist() == [0, 2]


def test_ties_keep_lower_anchor_first_and_max_det():
    n = 400
    boxes = [[i * 20, 0, i * 20 + 10, 10] for i in range(n)]  # disjoint
    p = _pred(boxes, [0.75] * n, [0] * n)
    _, _, _, src = ON.non_max_suppression(p, 0.25, max_det=300)
    assert src.tolist() == list(range(300))


def test_empty():
    p = _pred([[0, 0, 10, 10]], [0.1], [0])
    box, sc, cls, src = ON.non_max_suppression(p, 0.25)
    assert box.shape == (0, 4) and src.size == 0
