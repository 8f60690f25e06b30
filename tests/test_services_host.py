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


def test_fused_order(tmp_path):
    path = _clip(tmp_path, 31, 30.0)
    bus = R.InProcessBus()
    det = FakeDetector(lambda fid: [([2, 2, 20, 20], 0.9, 19)])
    y = services.YOLOPipeline(det, bus, _cfg(), results_dir=tmp_path / "yolo")
    s = services.SAM3Pipeline(None, bus, _cfg(), results_dir=tmp_path / "sam3", yolo_results_dir=tmp_path / "yolo")
    d = services.DINOv3Pipeline(FakeEmbedder(), bus, None, _cfg(), results_dir=tmp_path / "dino")
    fused = services.FusedFeatureService(y, s, d)
    _run(fused.start())
    _run(bus.publish("video.preprocessed", {"video_id": "z", "processed_path": str(path)}))
    assert [p[0] for p in bus.published] == ["video.preprocessed", "pipeline.yolo", "pipeline.sam3", "pipeline.dinov3"]
    sam = json.load(open(tmp_path / "sam3" / "z_sam3.json"))
    assert all(sg["mask_available"] for sg in sam["segmentations"])  # YOLO ran first: no race


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
    """FusedExtractor's stream assignment: at most two SAM passes in flight by default."""
    from lmx.pipeline import stream_plan

    assert stream_plan(4) == (4, 0, 1, [2, 3, 2, 3])               # bench batch: 64 frames, passes of 16
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
