"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product): CPU restatement of the tracking service's ByteTrack
association, written independently of lmx/services/tracking.py — table-of-arrays state instead of objects, numpy IoU, and
scipy.optimize.linear_sum_assignment in the place of lap.lapjv.

Follows services/tracking-service/app/tracker/bytetrack.py:76-160 (update), kalman.py:22-141 (7-state constant-velocity
filter; filterpy.kalman.KalmanFilter predict / update with the Joseph-form covariance), matching.py:12-173 (IoU, assignment,
threshold filter), track.py:59-112 (life cycle), main.py:159-213 (per-video driver and summaries).

PARITY UNPINNED: filterpy and lap are not installed and the reference ships no tracker tests or fixtures, so neither this file
nor the product can be run against the reference's own numbers; the two implementations are checked against each other and
against closed-form cases (tests/test_tracking.py)."""
import numpy as np
from scipy.optimize import linear_sum_assignment

T, C, L, D = 1, 2, 3, 4  # tentative, confirmed, lost, deleted (track.py:13-18)
NAMES = {T: "TENTATIVE", C: "CONFIRMED", L: "LOST", D: "DELETED"}


def iou(a, b):
    a, b = np.atleast_2d(a).astype(np.float64), np.atleast_2d(b).astype(np.float64)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0.0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-6)


def associate(det_boxes, trk_boxes, thr):
    """(pairs, unmatched detections, unmatched tracks), IoU-only cost (the service never has per-detection embeddings)."""
    nd, nt = len(det_boxes), len(trk_boxes)
    if nd == 0 or nt == 0:
        return [], list(range(nd)), list(range(nt))
    m = iou(np.array(det_boxes), np.array(trk_boxes))
    r, c = linear_sum_assignment(1.0 - m)
    pairs = [(int(i), int(j)) for i, j in zip(r, c) if m[i, j] >= thr]
    weak = [(int(i), int(j)) for i, j in zip(r, c) if m[i, j] < thr]
    # order of the unmatched lists (it decides the ids new tracks get): never assigned first, then the assigned-but-too-weak
    # pairs in row order (matching.py:161-167 appends them)
    return (pairs, [i for i in range(nd) if i not in set(r)] + [i for i, _ in weak],
            [j for j in range(nt) if j not in set(c)] + [j for _, j in weak])


def to_z(b):
    w, h = b[2] - b[0], b[3] - b[1]
    return np.array([b[0] + w / 2, b[1] + h / 2, w * h, w / (h + 1e-6)])


def to_box(x):
    s, r = max(1e-6, x[2]), max(1e-6, x[3])
    w = np.sqrt(s * r)
    h = s / (w + 1e-6)
    return np.array([x[0] - w / 2, x[1] - h / 2, x[0] + w / 2, x[1] + h / 2])


F = np.eye(7)
F[:3, 4:] = np.eye(3)
H = np.eye(4, 7)
Q = np.diag([1, 1, 1, 1, 0.01, 0.01, 0.0001])
R = np.diag([1, 1, 10, 10.0])
P0 = np.diag([10, 10, 10, 10, 1e4, 1e4, 1e4])


def track_video(by_frame, high=0.6, low=0.1, match=0.8):
    """by_frame: {frame: [(box xyxy, confidence)]} -> (rows (frame, id, box, confidence, state name) of confirmed tracks after
    every frame, summaries [(id, first frame, last frame, #frames, last confidence)] of the tracks with >= 3 hits at the end)."""
    tr = []  # dicts: id, box, conf, state, hits, tsu, frames, x, P
    next_id = 0
    rows = []

    def hit(t, box, conf, frame):
        t["box"], t["conf"] = np.array(box, np.float64), conf
        t["hits"] += 1
        t["tsu"] = 0
        t["frames"].append(frame)
        if t["state"] == T and t["hits"] >= 3:
            t["state"] = C
        elif t["state"] == L:
            t["state"] = C
        y = to_z(t["box"]) - H @ t["x"]
        S = H @ t["P"] @ H.T + R
        K = t["P"] @ H.T @ np.linalg.inv(S)
        t["x"] = t["x"] + K @ y
        A = np.eye(7) - K @ H
        t["P"] = A @ t["P"] @ A.T + K @ R @ K.T

    def miss(t):
        t["tsu"] += 1
        if t["state"] == C and t["tsu"] > 30:
            t["state"] = L
        elif t["state"] == T and t["tsu"] > 3:
            t["state"] = D
        elif t["state"] == L and t["tsu"] > 90:
            t["state"] = D

    def predict(t):
        if t["x"][6] + t["x"][2] <= 0:
            t["x"][6] = 0
        t["x"] = F @ t["x"]
        t["P"] = F @ t["P"] @ F.T + Q
        t["box"] = to_box(t["x"])

    for frame in sorted(by_frame):
        dets = by_frame[frame]
        if not dets:
            for t in tr:
                predict(t)
                miss(t)
        else:
            hi = [d for d in dets if d[1] >= high]
            lo = [d for d in dets if low <= d[1] < high]
            live = [t for t in tr if t["state"] != D]
            for t in tr:
                predict(t)
            p1, ud1, ut1 = associate([d[0] for d in hi], [t["box"] for t in live], match)
            for i, j in p1:
                hit(live[j], hi[i][0], hi[i][1], frame)
            left = [live[j] for j in ut1]
            p2, _, _ = associate([d[0] for d in lo], [t["box"] for t in left], 0.5)
            for i, j in p2:
                hit(left[j], lo[i][0], lo[i][1], frame)
            lost = [t for t in tr if t["state"] == L]
            rest = [hi[i] for i in ud1]
            p3, ud3, _ = associate([d[0] for d in rest], [t["box"] for t in lost], 0.3)
            for i, j in p3:
                hit(lost[j], rest[i][0], rest[i][1], frame)
            revived = {id(lost[j]) for _, j in p3}
            for t in left:  # including the ones the second stage just matched (bytetrack.py:147-149)
                if id(t) not in revived:
                    miss(t)
            for i in ud3:
                b = np.array(rest[i][0], np.float64)
                x = np.zeros(7)
                x[:4] = to_z(b)
                tr.append(dict(id=next_id, box=b, conf=rest[i][1], state=T, hits=1, tsu=0, frames=[frame], x=x, P=P0.copy()))
                next_id += 1
            tr = [t for t in tr if t["state"] != D]
            if len(tr) > 100:
                tr = sorted(tr, key=lambda t: t["tsu"])[:100]
        for t in tr:
            if t["state"] == C:
                rows.append((frame, t["id"], t["box"].copy(), t["conf"], NAMES[t["state"]]))
    return rows, [(t["id"], t["frames"][0], t["frames"][-1], len(t["frames"]), t["conf"]) for t in tr if t["hits"] >= 3]
