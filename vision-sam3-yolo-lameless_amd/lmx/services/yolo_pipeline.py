"""YOLOPipeline — mirror of services/yolo-pipeline/app/main.py: same sampling (every max(1, int(fps)//2)-th frame), same
per-detection dicts, same 11 aggregate features, same JSON file and `pipeline.yolo` payload.
The `self.yolo_model(frame, verbose=False, conf=...)` call (main.py:76) is replaced by one batched
``YoloDetector.detect`` over all sampled frames of the clip."""
import json
import traceback
from pathlib import Path

import numpy as np
import torch

from . import runtime as R


def compute_features(detections, total_frames, fps):
    """The 11 statistics of YOLOPipeline._compute_features (main.py:120-164), numpy float64 like the reference."""
    boxes = [d["bbox"] for fr in detections for d in fr["detections"]]
    confs = [d["confidence"] for fr in detections for d in fr["detections"]]
    if not detections or not boxes:
        return {}
    b = np.array(boxes)
    c = np.array(confs)
    wdt, hgt = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    cx, cy = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
    return {
        "num_detections": len(b),
        "avg_confidence": float(np.mean(c)),
        "max_confidence": float(np.max(c)),
        "min_confidence": float(np.min(c)),
        "avg_box_area": float(np.mean(wdt * hgt)),
        "avg_box_width": float(np.mean(wdt)),
        "avg_box_height": float(np.mean(hgt)),
        "position_stability": float(1.0 / (1.0 + np.std(cx) + np.std(cy))),
        "avg_center_x": float(np.mean(cx)),
        "avg_center_y": float(np.mean(cy)),
        "detection_rate": len(detections) / total_frames if total_frames > 0 else 0,
    }


def detections_from_device(frame_ids, fps, names, boxes, scores, cls, counts):
    """Device NMS output of the sampled frames -> the reference's list of per-frame dicts (frames without a detection
    are omitted, main.py:98-103).  One D2H copy per tensor for the whole clip instead of 3 syncs per box (:82-84)."""
    boxes, scores, cls, counts = (t.cpu().numpy() for t in (boxes, scores, cls, counts))
    out = []
    for j, fid in enumerate(frame_ids):
        dets = []
        for k in range(int(counts[j])):
            ci = int(cls[j, k])
            dets.append({"frame": fid, "bbox": [float(v) for v in boxes[j, k]], "confidence": float(scores[j, k]),
                         "class": names[ci] if ci in names else f"class_{ci}", "class_id": ci})
        if dets:
            out.append({"frame": fid, "time": fid / fps if fps > 0 else 0, "detections": dets})
    return out


class YOLOPipeline:
    def __init__(self, detector, bus, config=None, results_dir="/app/data/results/yolo", batch=32):
        self.config = config or R.load_config()
        self.nats_client = bus
        self.yolo_model = detector
        self.confidence_threshold = self.config.get("models", {}).get("yolo", {}).get("confidence_threshold", 0.5)
        self.results_dir = Path(results_dir)
        self.results_dir.mkdir(parents=True, exist_ok=True)
        self.batch = batch

    def results_from_detections(self, dets, total, fps):
        """The dict detect_in_video returns (main.py:109-118), from the per-frame detection dicts."""
        return {"detections": dets, "features": compute_features(dets, total, fps), "total_frames": total, "fps": fps,
                "frames_processed": len(dets)}

    def detect_in_video(self, video_path):
        clip = R.Clip.open(video_path)
        fps, total = clip.fps, clip.total_frames
        dets = []
        dev = self.yolo_model.device
        # the decode loop of main.py:63-107 as a stream: only the sampled frames are kept, `batch` at a time
        for chunk, host in clip.batches(max(1, fps // 2), self.batch):
            frames = torch.from_numpy(host).to(dev)
            b, s, c, _, n = self.yolo_model.detect(frames, conf=self.confidence_threshold)
            dets += detections_from_device(chunk, fps, self.yolo_model.names, b, s, c, n)
        return self.results_from_detections(dets, total, fps)

    async def write_and_publish(self, video_id, results):
        """main.py:181-199: `{id}_yolo.json` (indent 2) and the `pipeline.yolo` message."""
        results_file = self.results_dir / f"{video_id}_yolo.json"
        with open(results_file, "w") as f:
            json.dump(results, f, indent=2)
        await self.nats_client.publish(self.config["nats"]["subjects"]["pipeline_yolo"], {
            "video_id": video_id, "pipeline": "yolo", "results_path": str(results_file), "features": results["features"],
            "num_detections": len(results["detections"]), "total_frames": results["total_frames"]})

    async def process_video(self, video_data):
        video_id = video_data["video_id"]
        processed_path = Path(video_data["processed_path"])
        if not processed_path.exists():
            print(f"Processed video not found: {processed_path}")
            return
        try:
            await self.write_and_publish(video_id, self.detect_in_video(processed_path))
        except Exception as e:  # noqa: BLE001 — the reference never raises out of the handler (main.py:203-206)
            print(f"Error in YOLO pipeline for {video_id}: {e}")
            traceback.print_exc()

    async def start(self):
        await self.nats_client.connect()
        await self.nats_client.subscribe(self.config["nats"]["subjects"]["video_preprocessed"], self.process_video)
