"""SURVEY.md §8f rank 4 — the tracking service after the hot path (lmx/services/tracking.py + csrc/host_track.cpp) against
oracle/tracking.py (an independent restatement on scipy's assignment solver) and closed-form cases.  PARITY UNPINNED against
the reference's own dependencies (filterpy, lap: absent; the reference ships no tracker tests)."""
import asyncio
import json
import os

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from lmx.services import tracking as TR
from lmx.services.runtime import InProcessBus
from oracle import tracking as OT


def test_assignment_is_optimal_on_rectangular_costs():
    rng = np.random.default_rng(0)
    for n, m in [(1, 1), (3, 3), (5, 2), (2, 7), (17, 23), (40, 40), (64, 9)]:
        for _ in range(5):
            cost = rng.random((n, m))
            matched, ur, uc = TR.linear_assignment(cost)
            r, c = linear_sum_assignment(cost)
            assert len(matched) == min(n, m) and len(ur) == n - len(matched) and len(uc) == m - len(matched)
            assert len(set(matched[:, 0])) == len(matched) and len(set(matched[:, 1])) == len(matched)
            assert np.isclose(cost[matched[:, 0], matched[:, 1]].sum(), cost[r, c].sum(), rtol=0, atol=1e-12)
    m0, ur, uc = TR.linear_assignment(np.empty((0, 4)))
    assert m0.shape == (0, 2) and list(uc) == [0, 1, 2, 3] and len(ur) == 0
    with pytest.raises(Exception):
        TR.linear_assignment(np.array([[np.nan, 1.0]]))


def test_iou_matrix_matches_the_numpy_form():
    rng = np.random.default_rng(1)
    a = rng.random((9, 4)) * 100
    a[:, 2:] += a[:, :2]
    b = rng.random((6, 4)) * 100
    b[:, 2:] += b[:, :2]
    assert np.array_equal(TR.iou_batch(a, b), OT.iou(a, b))
    assert TR.iou_batch(np.array([0, 0, 10, 10.0]), np.array([[0, 0, 10, 10.0]]))[0, 0] == 100.0 / (100.0 + 1e-6)
    assert TR.iou_batch(np.array([[0, 0, 1, 1.0]]), np.array([[5, 5, 6, 6.0]]))[0, 0] == 0.0


def test_threshold_filter_returns_weak_matches_to_the_unmatched_lists():
    dets = np.array([[0, 0, 10, 10.0], [100, 100, 110, 110.0]])
    trks = np.array([[1, 1, 11, 11.0], [104, 104, 114, 114.0]])  # IoU 0.68 and 0.22
    m, ud, ut = TR.associate_detections_to_tracks(dets, trks, iou_threshold=0.3)
    assert m.tolist() == [[0, 0]] and ud.tolist() == [1] and ut.tolist() == [1]
    m, ud, ut = TR.associate_detections_to_tracks(dets, np.empty((0, 4)), 0.3)
    assert len(m) == 0 and ud.tolist() == [0, 1] and len(ut) == 0


def test_kalman_constant_velocity_box():
    """A box moving 5 px / frame at constant size: after a few updates the one-step prediction is within a pixel of the truth,
    and the covariance follows the Joseph form (symmetric, positive)."""
    k = TR.KalmanBoxTracker(np.array([0, 0, 20, 10.0]))
    for i in range(1, 12):
        pred = k.predict()
        truth = np.array([5.0 * i, 0, 5.0 * i + 20, 10.0])
        if i > 6:
            assert np.abs(pred - truth).max() < 1.0
        k.update(truth)
    assert np.allclose(k.P, k.P.T) and np.all(np.linalg.eigvalsh(k.P) > 0)
    assert k.hits == 11 and k.time_since_update == 0 and k.age == 11
    # a track whose area velocity would drive the area negative has that velocity zeroed before the step (kalman.py:124-125)
    k.x[2], k.x[6] = 4.0, -10.0
    k.predict()
    assert k.x[6] == 0 and k.x[2] == 4.0


def synth_sequence(seed, n_frames=120, n_obj=4):
    """Moving boxes with jitter, confidence dips into the low band, missed detections, an occlusion gap and clutter."""
    rng = np.random.default_rng(seed)
    pos = rng.random((n_obj, 2)) * [1500, 800] + [100, 100]
    vel = (rng.random((n_obj, 2)) - 0.5) * 16
    size = rng.random((n_obj, 2)) * [200, 120] + [150, 90]
    gap = {int(o): (int(rng.integers(20, 60)), int(rng.integers(5, 50))) for o in range(n_obj)}
    frames = {}
    for f in range(n_frames):
        dets = []
        for o in range(n_obj):
            c = pos[o] + vel[o] * f
            g0, gl = gap[o]
            if g0 <= f < g0 + gl and o % 2 == 0:
                continue
            if rng.random() < 0.08:
                continue
            j = rng.normal(0, 2.0, 4)
            box = [c[0] - size[o, 0] / 2 + j[0], c[1] - size[o, 1] / 2 + j[1], c[0] + size[o, 0] / 2 + j[2], c[1] + size[o, 1] / 2 + j[3]]
            conf = float(np.clip(rng.normal(0.8, 0.15), 0.05, 0.99)) if rng.random() > 0.15 else float(rng.uniform(0.12, 0.55))
            dets.append((box, conf))
        if rng.random() < 0.1:
            x, y = rng.random(2) * [1700, 900]
            dets.append(([x, y, x + 80, y + 60], float(rng.uniform(0.6, 0.9))))
        if f % 37 != 36:  # some frames have no detections at all
            frames[f * 15] = dets if f % 41 != 40 else []
    return frames


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_bytetrack_equals_the_independent_restatement(seed):
    frames = synth_sequence(seed)
    yolo = {"video_id": "v", "detections": [{"frame": f, "detections": [{"bbox": [float(v) for v in b], "confidence": c, "class_id": 19}
                                                                       for b, c in dets]} for f, dets in frames.items()]}
    by_frame, rows, summaries, stats = TR.track_video(yolo)
    o_rows, o_sum = OT.track_video(frames)
    assert len(rows) == len(o_rows) and len(rows) > 50
    for r, (f, tid, box, conf, state) in zip(rows, o_rows):
        assert (r["frame"], r["track_id"], r["state"]) == (f, tid, state)
        assert np.allclose(r["bbox"], box, rtol=0, atol=1e-6) and r["confidence"] == conf
    assert [(s["track_id"], s["start_frame"], s["end_frame"], s["total_frames"]) for s in summaries] == [s[:4] for s in o_sum]
    assert all(s["avg_confidence"] == pytest.approx(o[4]) for s, o in zip(summaries, o_sum))
    assert stats["total_tracks"] >= len(summaries) and stats["frame_id"] == max(frames) + 1
    assert len({r["track_id"] for r in rows}) >= 3


def test_life_cycle_and_the_second_stage_quirk():
    box = np.array([100, 100, 300, 250.0])
    t = TR.ByteTracker()
    for f in range(3):
        out = t.update([TR.Detection(box.copy(), 0.9)], f)
    assert [x.state for x in t.tracks] == [TR.CONFIRMED] and len(out) == 1 and t.tracks[0].hits == 3
    # a low-confidence detection keeps the track alive through the SECOND stage, and the reference then marks it missed anyway
    t.update([TR.Detection(box.copy(), 0.3)], 3)
    assert t.tracks[0].hits == 4 and t.tracks[0].time_since_update == 1
    # nothing for 31 frames: CONFIRMED -> LOST; a far-away detection in between starts its own tentative track and dies
    t.update([TR.Detection(np.array([1500, 800, 1600, 900.0]), 0.9)], 4)
    for f in range(5, 36):
        t.update([], f)
    states = {x.track_id: x.state for x in t.tracks}
    assert states[0] == TR.LOST and states.get(1, TR.DELETED) == TR.DELETED
    # a detection below the high threshold does not start a track; one above it near the lost box revives track 0
    n_before = t.next_id
    t.update([TR.Detection(np.array([900, 100, 1000, 200.0]), 0.5)], 36)
    assert t.next_id == n_before
    revived = t.update([TR.Detection(t.tracks[0].bbox.copy(), 0.95)], 37)
    assert [x.track_id for x in revived] == [0] and t.tracks[0].state == TR.CONFIRMED


def test_parse_formats():
    d = {"bbox": [0, 0, 1, 1], "confidence": 0.9}
    assert TR.parse_yolo_detections({"detections": [{"frame": 3, "detections": [d, d]}, {"frame": 4, "bbox": [0, 0, 1, 1], "confidence": 0.5}]}) == \
        {3: [d, d], 4: [{"frame": 4, "bbox": [0, 0, 1, 1], "confidence": 0.5}]}
    assert TR.parse_yolo_detections({"frames": [{"frame_number": 7, "detections": [d]}]}) == {7: [d]}
    assert TR.parse_yolo_detections({"frame_results": {"9": [d]}}) == {9: [d]}
    assert TR.parse_yolo_detections({}) == {}


def test_reid_thresholds_and_momentum():
    store = TR.MemoryIdentityStore()
    ids = iter(["id-a", "id-b", "id-c"])
    m = TR.CowReIDMatcher(store, embedding_dim=8, new_uuid=lambda: next(ids))
    m.connect()
    e = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
    r1 = m.match_or_create(e * 3, "v1", 0)
    assert r1 == {"identity_id": "id-a", "cow_id": "COW-0001", "similarity": 1.0, "confidence": "high", "is_new_identity": True}
    near = np.array([0.9, np.sqrt(1 - 0.81), 0, 0, 0, 0, 0, 0])  # cosine 0.9 -> high, identity vector moves towards it
    r2 = m.match_or_create(near, "v2", 1)
    assert r2["cow_id"] == "COW-0001" and r2["confidence"] == "high" and not r2["is_new_identity"] and r2["similarity"] == pytest.approx(0.9)
    v, payload = store.retrieve("id-a")
    assert payload["total_sightings"] == 2 and np.linalg.norm(v) == pytest.approx(1.0)
    want = 0.9 * e + 0.1 * near
    assert np.allclose(v, want / np.linalg.norm(want))
    mid = np.array([0.7, np.sqrt(1 - 0.49), 0, 0, 0, 0, 0, 0])  # ~0.75 against the moved vector is not guaranteed: use a clear miss
    far = np.array([0.0, 0, 1, 0, 0, 0, 0, 0])
    r3 = m.match_or_create(far, "v3", 2)
    assert r3["cow_id"] == "COW-0002" and r3["is_new_identity"] and store.count() == 2
    best, cands = m.match_embedding(mid)
    assert cands[0]["cow_id"] == "COW-0001" and cands[0]["confidence"] in ("low", "medium") and len(cands) == 2


def test_service_files_and_subjects(tmp_path):
    frames = synth_sequence(7, n_frames=40, n_obj=2)
    yolo = {"video_id": "clip7", "pipeline": "yolo",
            "detections": [{"frame": f, "time": f / 30, "detections": [{"bbox": [float(v) for v in b], "confidence": c, "class": "cow", "class_id": 19}
                                                                      for b, c in dets]} for f, dets in frames.items()]}
    yolo_path = tmp_path / "clip7_yolo.json"
    yolo_path.write_text(json.dumps(yolo))
    dino_path = tmp_path / "clip7_dinov3.json"
    emb = np.random.default_rng(0).normal(size=16)
    dino_path.write_text(json.dumps({"video_id": "clip7", "canonical_frames": [{"frame": 0, "embedding": emb.tolist()},
                                                                               {"frame": 30, "embedding": (emb * 3).tolist()}]}))
    bus = InProcessBus()
    got = []

    async def run():
        svc = TR.TrackingService(bus, TR.MemoryIdentityStore(), results_dir=str(tmp_path / "tracking"))
        await svc.start()
        await bus.subscribe("tracking.complete", lambda m: got.append(("complete", m)))
        await bus.subscribe("tracking.reid.match", lambda m: got.append(("reid", m)))
        await bus.publish("pipeline.yolo", {"video_id": "clip7", "results_path": str(yolo_path)})
        await bus.publish("pipeline.dinov3", {"video_id": "clip7", "results_path": str(dino_path)})
        await bus.publish("pipeline.yolo", {"video_id": "nofile", "results_path": str(tmp_path / "missing.json")})
        return svc

    svc = asyncio.run(run())
    assert [g[0] for g in got] == ["complete", "reid"]
    res = json.loads((tmp_path / "tracking" / "clip7_tracking.json").read_text())
    assert list(res.keys()) == ["video_id", "pipeline", "total_tracks", "track_summaries", "frame_tracks", "statistics", "reid_results", "reid_complete"]
    assert res["pipeline"] == "tracking" and res["total_tracks"] == len(res["track_summaries"]) >= 2 and res["reid_complete"] is True
    assert list(res["frame_tracks"][0].keys()) == ["frame", "track_id", "bbox", "confidence", "state"]
    assert list(res["track_summaries"][0].keys()) == ["track_id", "start_frame", "end_frame", "total_frames", "avg_confidence"]
    assert got[0][1] == {"video_id": "clip7", "results_path": os.path.join(str(tmp_path / "tracking"), "clip7_tracking.json"),
                         "total_tracks": res["total_tracks"], "pending_reid": True}
    # the clip's mean canonical embedding stands for every track: the first creates COW-0001, the others match it at ~1.0
    cows = [r["cow_id"] for r in res["reid_results"]]
    assert cows == ["COW-0001"] * len(cows) and [r["is_new"] for r in res["reid_results"]] == [True] + [False] * (len(cows) - 1)
    assert got[1][1]["new_identities"] == 1 and "clip7" not in svc.pending_tracks
    assert np.allclose(svc.video_embeddings["clip7"], emb * 2)
