"""BASELINE cfg#1: services/yolo-pipeline on ONE 640 x 640 frame, presented as a 1-frame clip (fps 30,
total_frames 1) through the restated `process_video` (yolo main.py:166-206), YOLOv8-n, conf 0.001 so that NMS and the
max_det = 300 cut are exercised; JSON schema and publish payload = Appendix B.1.
  * CPU (`-m "not gpu"`): the fp32 oracle behind the detector seam (tests/oracle_backend.py) — the plumbing of the config
    as BASELINE.json words it ("CPU PyTorch reference path (plumbing, no GPU)"), and the committed golden detections;
  * GPU: the HIP detector (exact plan) through the SAME service code on the SAME clip: the JSON carries the fp32 oracle's
    detections — identical count, order and classes (keep-set = golden `src`), confidences within 2e-5, boxes within 5e-3 px."""
import asyncio
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FEATURE_KEYS = ["num_detections", "avg_confidence", "max_confidence", "min_confidence", "avg_box_area", "avg_box_width",
                "avg_box_height", "position_stability", "avg_center_x", "avg_center_y", "detection_rate"]


def _service(tmp_path, detector, tag):
    from lmx import services, synth
    from lmx.services import runtime as R

    clip = tmp_path / "one_frame.npz"
    if not clip.exists():
        R.save_npz_clip(clip, synth.cfg1_frame()[None], 30.0)
    bus = R.InProcessBus()
    cfg = {"nats": {"subjects": dict(R.DEFAULT_SUBJECTS)}, "models": {"yolo": {"confidence_threshold": 0.001}}}
    svc = services.YOLOPipeline(detector, bus, cfg, results_dir=tmp_path / tag)
    asyncio.run(svc.start())
    asyncio.run(bus.publish("video.preprocessed", {"video_id": "cfg1", "processed_path": str(clip)}))
    return json.load(open(tmp_path / tag / "cfg1_yolo.json")), bus.published


def _check_schema(res, published, names):
    assert list(res) == ["detections", "features", "total_frames", "fps", "frames_processed"]
    assert res["total_frames"] == 1 and res["fps"] == 30 and res["frames_processed"] == 1
    assert len(res["detections"]) == 1
    fr = res["detections"][0]
    assert list(fr) == ["frame", "time", "detections"] and fr["frame"] == 0 and fr["time"] == 0.0
    for d in fr["detections"]:
        assert list(d) == ["frame", "bbox", "confidence", "class", "class_id"]
        assert d["frame"] == 0 and len(d["bbox"]) == 4 and all(isinstance(v, float) for v in d["bbox"])
        assert isinstance(d["confidence"], float) and isinstance(d["class_id"], int) and d["class"] == names[d["class_id"]]
        assert 0.0 <= d["bbox"][0] <= d["bbox"][2] <= 640.0 and 0.0 <= d["bbox"][1] <= d["bbox"][3] <= 640.0
    assert list(res["features"]) == FEATURE_KEYS
    assert res["features"]["num_detections"] == len(fr["detections"]) and res["features"]["detection_rate"] == 1.0
    subj, payload = published[-1]
    assert subj == "pipeline.yolo"
    assert list(payload) == ["video_id", "pipeline", "results_path", "features", "num_detections", "total_frames"]
    assert payload["video_id"] == "cfg1" and payload["pipeline"] == "yolo" and payload["num_detections"] == 1
    assert payload["features"] == res["features"] and payload["total_frames"] == 1
    confs = [d["confidence"] for d in fr["detections"]]
    assert confs == sorted(confs, reverse=True) and confs[-1] > 0.001
    return fr["detections"]


def _model():
    from lmx import yolo

    cfg = yolo.YoloConfig("n")
    return cfg, yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("n"))


def test_cfg1_cpu_reference_path(tmp_path):
    from lmx import yolo
    from oracle_backend import OracleDetector

    cfg, sd = _model()
    names = {i: n for i, n in enumerate(yolo.COCO_NAMES)}
    det = OracleDetector("n", cfg.nc, sd, names)
    res, published = _service(tmp_path, det, "cpu")
    dets = _check_schema(res, published, names)
    g = np.load(os.path.join(GOLD, "cfg1_yolov8n_w7.npz"))
    assert len(dets) == len(g["src"]) == 300, "conf 0.001 must reach the max_det = 300 cut"
    assert np.array_equal(det.src[0], g["src"]) and [d["class_id"] for d in dets] == g["cls"].tolist()
    assert np.allclose([d["confidence"] for d in dets], g["scores"], atol=2e-6)  # (another core count reorders the fp32 sums)
    assert np.allclose([d["bbox"] for d in dets], g["boxes"], atol=2e-3)


@pytest.mark.gpu
def test_cfg1_hip_backend_carries_the_fp32_detections(cuda, tmp_path):
    from lmx import synth, yolo

    cfg, sd = _model()
    det = yolo.YoloDetector(cfg, sd, cuda)  # exact plan
    res, published = _service(tmp_path, det, "hip")
    dets = _check_schema(res, published, det.names)
    g = np.load(os.path.join(GOLD, "cfg1_yolov8n_w7.npz"))
    assert len(dets) == len(g["src"])
    assert [d["class_id"] for d in dets] == g["cls"].tolist(), "classes / order differ from the fp32 oracle's detections"
    ds = float(np.abs(np.asarray([d["confidence"] for d in dets]) - g["scores"]).max())
    db = float(np.abs(np.asarray([d["bbox"] for d in dets]) - g["boxes"]).max())
    # the keep-set itself (anchor indices are not part of the JSON): the same call the service made
    src, counts = (t.cpu().numpy() for t in det.detect(torch.from_numpy(synth.cfg1_frame()[None]).to(cuda), conf=0.001)[3:5])
    assert int(counts[0]) == len(g["src"]) and np.array_equal(src[0, :counts[0]], g["src"]), "keep-set differs from the fp32 golden"
    print(f"cfg#1: {len(dets)} detections = fp32 oracle's, confidence within {ds:.2e}, boxes within {db:.2e} px")
    assert ds <= 2e-5 and db <= 5e-3
