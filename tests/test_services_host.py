"""Service contracts with stub backends on the CPU: subjects, JSON keys, sampling, quirks and
error conventions of services/{yolo,sam3,dinov3}-pipeline/app/main.py.  The model call is faked here; the GPU test
tests/test_gpu_services.py runs the same services over the real HIP backends."""
import asyncio
import json

import numpy as np
import pytest
import torch

from lmx import services
from lmx.services import runtime as R
from lmx.services import yolo_pipeline as YP


class FakeDetector:
    device = torch.device("cpu")
    names = {0: "person", 19: "cow"}

    def __init__(self, per_frame):
        self.per_frame = per_frame  # frame mean value -> list of (box, score, cls)
        self.calls = []

    def detect(self, frames, conf=0.25, iou=0.7, max_det=300):
        n = frames.shape[0]
        self.calls.append((n, conf))
        boxes = torch.zeros((n, 300, 4))
        scores = torch.zeros((n, 300))
        cls = torch.zeros((n, 300), dtype=torch.int32)
        counts = torch.zeros((n,), dtype=torch.int32)
        for j in range(n):
            dets = self.per_frame(int(frames[j, 0, 0, 0]))
            for k, (b, s, c) in enumerate(dets):
                boxes[j, k] = torch.tensor(b)
                scores[j, k], cls[j, k] = s, c
            counts[j] = len(dets)
        return boxes, scores, cls, None, counts


class FakeEmbedder:
    device = torch.device("cpu")

    class cfg:
        hidden = 8

    def embed_frames(self, frames):
        base = frames[:, 0, 0, 0].float()
        return torch.stack([base + i for i in range(8)], 1) / 10


def _clip(tmp_path, n, fps, h=48, w=64):
    frames = np.zeros((n, h, w, 3), np.uint8)
    frames[:, 0, 0, 0] = np.arange(n)  # frame index in the first byte
    p = tmp_path / "clip.npz"
    R.save_npz_clip(p, frames, fps)
    return p


def _cfg():
    return {"nats": {"subjects": dict(R.DEFAULT_SUBJECTS)}, "models": {"yolo": {"confidence_threshold": 0.5}}}


def _run(coro):
    return asyncio.run(coro)


def test_yolo_schema_sampling_and_quirks(tmp_path):
    path = _clip(tmp_path, 45, 29.97)  # int(fps) = 29 -> interval 14 -> frames 0,14,28,42
    det = FakeDetector(lambda fid: [([1.5, 2.0, 30.25, 40.0], 0.9, 19), ([5, 5, 10, 10], 0.6, 0)] if fid in (0, 28) else [])
    bus = R.InProcessBus()
    svc = services.YOLOPipeline(det, bus, _cfg(), results_dir=tmp_path / "yolo")
    _run(svc.process_video({"video_id": "v1", "processed_path": str(path)}))
    res = json.load(open(tmp_path / "yolo" / "v1_yolo.json"))
    assert list(res) == ["detections", "features", "total_frames", "fps", "frames_processed"]
    assert res["fps"] == 29 and res["total_frames"] == 45
    assert [d["frame"] for d in res["detections"]] == [0, 28] and res["frames_processed"] == 2  # only frames WITH detections
    d0 = res["detections"][1]["detections"][0]
    assert list(d0) == ["frame", "bbox", "confidence", "class", "class_id"]
    assert d0["class"] == "cow" and d0["class_id"] == 19 and d0["bbox"] == [1.5, 2.0, 30.25, 40.0]
    assert res["detections"][1]["time"] == 28 / 29
    assert list(res["features"]) == ["num_detections", "avg_confidence", "max_confidence", "min_confidence", "avg_box_area",
                                     "avg_box_width", "avg_box_height", "position_stability", "avg_center_x", "avg_center_y",
                                     "detection_rate"]
    assert res["features"]["num_detections"] == 4 and res["features"]["detection_rate"] == 2 / 45
    assert det.calls == [(4, 0.5)]  # ONE batched call with the configured threshold
    subj, payload = bus.published[0]
    assert subj == "pipeline.yolo" and list(payload) == ["video_id", "pipeline", "results_path", "features", "num_detections", "total_frames"]
    assert payload["num_detections"] == 2  # = number of frames with detections (Appendix B.1)


def test_yolo_feature_formulas():
    dets = [{"frame": 0, "time": 0, "detections": [{"bbox": [0, 0, 10, 20], "confidence": 0.5}, {"bbox": [10, 10, 30, 20], "confidence": 1.0}]}]
    f = YP.compute_features(dets, 10, 30)
    assert f["avg_box_area"] == 200.0 and f["avg_box_width"] == 15.0 and f["avg_center_x"] == 12.5
    assert f["position_stability"] == pytest.approx(1 / (1 + np.std([5, 20]) + np.std([10, 15])))
    assert YP.compute_features([], 10, 30) == {}


def test_error_conventions(tmp_path, capsys):
    bus = R.InProcessBus()
    svc = services.YOLOPipeline(FakeDetector(lambda f: []), bus, _cfg(), results_dir=tmp_path / "y")
    _run(svc.start())
    _run(bus.publish("video.preprocessed", {"video_id": "x"}))  # admin single trigger: no processed_path -> KeyError swallowed
    assert "Error processing message" in capsys.readouterr().out
    _run(svc.process_video({"video_id": "x", "processed_path": str(tmp_path / "missing.npz")}))
    assert not (tmp_path / "y" / "x_yolo.json").exists() and len(bus.published) == 1
    bad = tmp_path / "bad.npz"
    bad.write_bytes(b"not a clip")
    _run(svc.process_video({"video_id": "x", "processed_path": str(bad)}))  # exception inside: print, no file, no publish
    assert not (tmp_path / "y" / "x_yolo.json").exists() and len(bus.published) == 1


def test_sam3_rectangle_fallback_and_missing_yolo(tmp_path):
    path = _clip(tmp_path, 31, 30.0)  # interval 15 -> 0, 15, 30
    bus = R.InProcessBus()
    ydir = tmp_path / "yolo"
    ydir.mkdir()
    svc = services.SAM3Pipeline(None, bus, _cfg(), results_dir=tmp_path / "sam3", yolo_results_dir=ydir)
    _run(svc.process_video({"video_id": "v", "processed_path": str(path)}))  # the race: YOLO file not there yet
    res = json.load(open(tmp_path / "sam3" / "v_sam3.json"))
    assert [s["mask_available"] for s in res["segmentations"]] == [False, False, False] and res["aggregated_features"] == {}
    json.dump({"detections": [{"frame": 15, "time": 0.5, "detections": [{"bbox": [10.9, 5.2, 30.7, 25.9]}, {"bbox": [0, 0, 5, 5]}]}]},
              open(ydir / "w_yolo.json", "w"))
    _run(svc.process_video({"video_id": "w", "processed_path": str(path)}))
    res = json.load(open(tmp_path / "sam3" / "w_sam3.json"))
    assert list(res) == ["segmentations", "aggregated_features", "total_frames", "fps", "frames_processed"]
    seg = res["segmentations"][1]
    assert seg["mask_available"] and list(seg["features"]) == ["mask_area", "area_ratio", "circularity", "aspect_ratio",
                                                                 "centroid_x", "centroid_y", "perimeter", "frame", "time"]
    assert seg["features"]["mask_area"] == 20 * 20 and seg["features"]["aspect_ratio"] == 1.0  # int() truncation: [10:30, 5:25]
    assert seg["features"]["perimeter"] == 76 and seg["features"]["centroid_x"] == 19.5
    assert res["frames_processed"] == 3 and list(res["aggregated_features"]) == ["avg_mask_area", "avg_area_ratio", "avg_circularity", "avg_aspect_ratio"]
    assert bus.published[-1][0] == "pipeline.sam3" and bus.published[-1][1]["num_segmentations"] == 3


def test_dinov3_schema_and_neighbor_evidence(tmp_path):
    path = _clip(tmp_path, 125, 25.0)  # one frame per second: 0,25,50,75,100
    bus, store = R.InProcessBus(), R.MemoryVectorStore()
    store.upsert("a", [1, 2, 3, 4, 5, 6, 7, 8], {"video_id": "a", "label": 1})
    store.upsert("b", [8, 7, 6, 5, 4, 3, 2, 1], {"video_id": "b", "label": 0})
    store.upsert("c", [1, 1, 1, 1, 1, 1, 1, 1], {"video_id": "c", "label": None})
    svc = services.DINOv3Pipeline(FakeEmbedder(), bus, store, _cfg(), results_dir=tmp_path / "d")
    _run(svc.process_video({"video_id": "v", "processed_path": str(path), "filename": "f.mp4"}))
    res = json.load(open(tmp_path / "d" / "v_dinov3.json"))
    assert list(res) == ["video_id", "embedding_dim", "num_embeddings", "similar_cases", "neighbor_evidence", "canonical_frames"]
    assert res["embedding_dim"] == 8 and res["num_embeddings"] == 5
    assert [c["frame"] for c in res["canonical_frames"]] == [0, 50, 100]
    assert res["neighbor_evidence"] == 0.5 and len(res["similar_cases"]) == 3  # labels [1, 0] among the neighbours
    assert "v" in store.points and store.points["v"][1]["filename"] == "f.mp4"
    subj, payload = bus.published[-1]
    assert subj == "pipeline.dinov3" and list(payload) == ["video_id", "pipeline", "results_path", "neighbor_evidence", "similar_cases", "embedding_dim"]


class FakeExtractor:
    """CPU stand-in for lmx.pipeline.FusedExtractor: FakeDetector boxes, the rectangle of the first box as the mask (what
    SAM3Pipeline(None) falls back to), FakeEmbedder embeddings; same step() surface and record fields."""
    device = torch.device("cpu")

    def __init__(self, det, emb):
        self.det, self.emb = det, emb
        self.yolo = det
        self.steps = []

    class dino:
        class cfg:
            hidden = 8

    def step(self, frames, conf=0.5, det_idx=None, emb_idx=None, **_):
        from lmx.services.sam3_pipeline import fallback_segmentation

        n, h, w, _c = frames.shape
        di = list(range(n)) if det_idx is None else list(det_idx)
        ei = list(range(n)) if emb_idx is None else list(emb_idx)
        self.steps.append((n, len(di), len(ei)))
        out = dict(boxes=torch.zeros((n, 300, 4)), scores=torch.zeros((n, 300)), cls=torch.zeros((n, 300), dtype=torch.int32),
                   counts=torch.zeros((n,), dtype=torch.int32), embedding=torch.zeros((n, 8)),
                   mask_bits=torch.zeros((n, h, (w + 7) // 8), dtype=torch.uint8), mask_stats=torch.zeros((n, 8), dtype=torch.int64),
                   mask_iou=torch.zeros((n,)), ran_det=torch.zeros((n,), dtype=torch.int32), ran_emb=torch.zeros((n,), dtype=torch.int32))
        if di:
            b, s, c, _x, k = self.det.detect(frames[di], conf=conf)
            for jj, j in enumerate(di):
                out["boxes"][j], out["scores"][j], out["cls"][j], out["counts"][j] = b[jj], s[jj], c[jj], k[jj]
                if int(k[jj]):
                    m = fallback_segmentation((h, w), b[jj, 0].tolist())
                    out["mask_bits"][j] = torch.from_numpy(np.packbits(m, axis=-1))
                out["ran_det"][j] = 1
        if ei:
            e = self.emb.embed_frames(frames[ei])
            for jj, j in enumerate(ei):
                out["embedding"][j] = e[jj]
                out["ran_emb"][j] = 1
        return out


def _three_and_fused(tmp_path, tag, bus, per_frame, schedule="reference", chunk=4):
    det = FakeDetector(per_frame)
    y = services.YOLOPipeline(det, bus, _cfg(), results_dir=tmp_path / tag / "yolo")
    s = services.SAM3Pipeline(None, bus, _cfg(), results_dir=tmp_path / tag / "sam3", yolo_results_dir=tmp_path / tag / "yolo")
    d = services.DINOv3Pipeline(FakeEmbedder(), bus, None, _cfg(), results_dir=tmp_path / tag / "dino")
    fx = FakeExtractor(FakeDetector(per_frame), FakeEmbedder())
    return y, s, d, services.FusedFeatureService(fx, y, s, d, schedule=schedule, chunk=chunk), fx


def _load3(root, vid):
    return [json.load(open(root / n / f"{vid}_{k}.json")) for n, k in (("yolo", "yolo"), ("sam3", "sam3"), ("dino", "dinov3"))]


@pytest.mark.parametrize("fps,n_frames,schedule", [(30.0, 95, "reference"), (25.0, 60, "reference"), (30.0, 47, "dense")])
def test_fused_service_equals_the_three_services(tmp_path, monkeypatch, fps, n_frames, schedule):
    """One open + one decode pass, three JSONs identical to what the three services write (fps 25: the DINO schedule
    25, 50 is NOT a subset of the YOLO/SAM schedule 0, 12, 24, ...), subjects in the order yolo -> sam3 -> dinov3."""
    path = _clip(tmp_path, n_frames, fps)
    per_frame = (lambda fid: [([2 + fid % 5, 2, 20 + fid % 7, 22], 0.9, 19), ([1, 1, 5, 5], 0.7, 0)] if fid not in (12, 30, 60) else [])
    bus_a, bus_b = R.InProcessBus(), R.InProcessBus()
    y, s, d, _, _ = _three_and_fused(tmp_path, "sep", bus_a, per_frame)
    msg = {"video_id": "z", "processed_path": str(path), "filename": "z.mp4"}
    for svc in (y, s, d):
        _run(svc.process_video(msg))
    _, _, _, fused, fx = _three_and_fused(tmp_path, "fused", bus_b, per_frame, schedule)
    opens, passes = [], []
    real_open, real_iter = R.Clip.open, R.Clip.iter_frames
    monkeypatch.setattr(R.Clip, "open", staticmethod(lambda p: (opens.append(p), real_open(p))[1]))
    monkeypatch.setattr(R.Clip, "iter_frames", lambda self, keep=None: (passes.append(1), real_iter(self, keep))[1])
    _run(fused.start())
    _run(bus_b.publish("video.preprocessed", msg))
    assert len(opens) == 1 and len(passes) == 1, "the fused service must open and decode the clip once"
    assert [p[0] for p in bus_b.published] == ["video.preprocessed", "pipeline.yolo", "pipeline.sam3", "pipeline.dinov3"]
    sep, fu = _load3(tmp_path / "sep", "z"), _load3(tmp_path / "fused", "z")
    assert sep == fu
    for (sa, pa), (sb, pb) in zip(bus_a.published, bus_b.published[1:]):
        assert sa == sb and {k: v for k, v in pa.items() if k != "results_path"} == {k: v for k, v in pb.items() if k != "results_path"}
    if schedule == "reference":  # every chunk carries only scheduled frames, and DINO runs on its own (sparser) schedule
        i_det, i_emb = max(1, int(fps) // 2), max(1, int(fps))
        union = sorted(set(range(0, n_frames, i_det)) | set(range(0, n_frames, i_emb)))
        assert sum(n for n, _, _ in fx.steps) == len(union)
        assert sum(k for _, k, _ in fx.steps) == len(range(0, n_frames, i_det)) and sum(k for _, _, k in fx.steps) == len(range(0, n_frames, i_emb))
    else:
        assert sum(n for n, _, _ in fx.steps) == n_frames
    sam = fu[1]
    assert any(sg["mask_available"] for sg in sam["segmentations"]) and not all(sg["mask_available"] for sg in sam["segmentations"])


def test_tracking_service_consumes_what_the_fused_service_publishes(tmp_path):
    """video.preprocessed -> (fused: pipeline.yolo, pipeline.sam3, pipeline.dinov3) -> tracking.complete -> tracking.reid.match on
    one bus: the tracker reads the YOLO JSON's per-frame containers and the DINO JSON's canonical_frames as they are written."""
    from lmx.services import tracking as TR

    path = _clip(tmp_path, 120, 30.0)
    per_frame = (lambda fid: [([4 + fid // 6, 6, 40 + fid // 6, 40], 0.9, 19), ([50, 2, 62, 12], 0.3, 0)])
    bus = R.InProcessBus()
    _, _, _, fused, _ = _three_and_fused(tmp_path, "f", bus, per_frame)
    trk = TR.TrackingService(bus, TR.MemoryIdentityStore(), _cfg(), results_dir=str(tmp_path / "tracking"))
    _run(fused.start())
    _run(trk.start())
    _run(bus.publish("video.preprocessed", {"video_id": "z", "processed_path": str(path), "filename": "z.mp4"}))
    assert [p[0] for p in bus.published] == ["video.preprocessed", "pipeline.yolo", "tracking.complete", "pipeline.sam3", "pipeline.dinov3",
                                            "tracking.reid.match"]
    res = json.load(open(tmp_path / "tracking" / "z_tracking.json"))
    assert res["total_tracks"] == 1 and res["track_summaries"][0]["total_frames"] == 8 and res["reid_complete"] is True
    assert [r["frame"] for r in res["frame_tracks"]] == [30, 45, 60, 75, 90, 105] and {r["track_id"] for r in res["frame_tracks"]} == {0}
    assert res["reid_results"][0]["cow_id"] == "COW-0001" and res["statistics"]["confirmed"] == 1


def test_clip_streams_and_keeps_only_sampled_frames(tmp_path, monkeypatch):
    """A video file is read like the reference's cap.read() loop: one decoded frame alive at a time, only sampled frames kept
    (the round-1 reader stacked every frame of the clip: 56 GB for five minutes of 1080p)."""
    import sys
    import types
    import weakref

    alive, peak = [0], [0]

    def _gone():
        alive[0] -= 1

    class Frame(np.ndarray):
        pass

    class Cap:
        def __init__(self, path):
            self.i, self.n = 0, 301

        def isOpened(self):
            return True

        def get(self, prop):
            return {5: 29.97, 7: 300, 3: 64, 4: 48}[prop]  # metadata says 300, the stream holds 301 (cv2 does that)

        def read(self):
            if self.i >= self.n:
                return False, None
            f = np.zeros((48, 64, 3), np.uint8).view(Frame)
            f[0, 0, 0] = self.i % 256
            alive[0] += 1
            weakref.finalize(f, _gone)
            peak[0] = max(peak[0], alive[0])
            self.i += 1
            return True, f

        def release(self):
            pass

    cv2 = types.SimpleNamespace(VideoCapture=Cap, CAP_PROP_FPS=5, CAP_PROP_FRAME_COUNT=7, CAP_PROP_FRAME_WIDTH=3, CAP_PROP_FRAME_HEIGHT=4)
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    clip = R.Clip.open(str(tmp_path / "video.mp4"))
    assert (clip.fps, clip.total_frames, clip.frame_hw) == (29, 300, (48, 64))
    got = []
    for ids, frames in clip.batches((14, 29), 8):
        assert frames.dtype == np.uint8 and frames.shape[1:] == (48, 64, 3) and len(ids) <= 8
        got += ids
        assert [int(v) for v in frames[:, 0, 0, 0]] == [i % 256 for i in ids]
    want = sorted(set(range(0, 301, 14)) | set(range(0, 301, 29)))
    assert got == want and clip.n_decoded == 301
    assert peak[0] <= 8 + 2, f"{peak[0]} decoded frames were alive at once"


def test_clip_curation_best_detection_rule():
    """services/clip-curation/app/main.py:106-131: largest box that is a cow (COCO 19) or covers > 10 % of the frame."""
    import numpy as np

    from lmx.services.curation import best_detection

    boxes = np.asarray([[0, 0, 100, 100], [10, 10, 500, 400], [0, 0, 1000, 700], [5, 5, 50, 60]], np.float32)
    scores = np.asarray([0.9, 0.8, 0.4, 0.95], np.float32)
    cls = np.asarray([19, 19, 3, 19], np.int32)
    d = best_detection(boxes, scores, cls, 4, 1080, 1920)   # the non-cow box is 33 % of the frame and the largest
    assert d["bbox"] == [0.0, 0.0, 1000.0, 700.0] and abs(d["confidence"] - 0.4) < 1e-6 and d["area"] == 700000.0
    assert d["centroid"] == (500.0, 350.0)
    d = best_detection(boxes, scores, cls, 2, 1080, 1920)   # only the first two rows are valid detections
    assert d["bbox"] == [10.0, 10.0, 500.0, 400.0]
    cls2 = np.asarray([3, 3, 3, 3], np.int32)
    assert best_detection(boxes[:2], scores, cls2, 2, 1080, 1920) is None  # small non-cow boxes are ignored
    assert best_detection(boxes, scores, cls, 0, 1080, 1920) is None


def test_stream_plan_layouts():
    """FusedExtractor's stream assignment (the product default is max_streams = 6: four SAM passes in flight)."""
    from lmx.pipeline import stream_plan

    assert stream_plan(4) == (4, 0, 1, [2, 3, 2, 3])               # 64 frames, passes of 16, at most 4 streams
    assert stream_plan(2) == (4, 0, 1, [2, 3])
    assert stream_plan(1) == (3, 0, 1, [2])
    assert stream_plan(4, 6) == (6, 0, 1, [2, 3, 4, 5])
    assert stream_plan(4, 1) == (3, 0, 1, [2, 2, 2, 2])            # never fewer than YOLO, DINO and one SAM stream
    assert stream_plan(4, 3, "rr") == (3, 0, 1, [2, 0, 1, 2])
    for n in range(1, 9):
        for cap in range(1, 8):
            k, d, e, sam = stream_plan(n, cap)
            assert len(sam) == n and d != e and all(2 <= s < k for s in sam) and len(set(sam)) <= max(1, min(cap, 2 + n) - 2)
    try:
        stream_plan(2, 4, "spiral")
    except ValueError:
        pass
    else:
        raise AssertionError("an unknown layout must be refused")
