"""The fp32 CPU oracle behind the detector seam of the service mirrors — TEST INFRASTRUCTURE (SURVEY.md section 8d's `cpu_ref`
backend): `detect(frames, conf=...)` with YoloDetector's return convention, one oracle.yolo.predict call per frame, like the
reference's per-frame `self.yolo_model(frame, verbose=False, conf=...)` loop (yolo main.py:69-105)."""
import numpy as np
import torch

from oracle import yolo as OY


class OracleDetector:
    device = torch.device("cpu")

    def __init__(self, scale, nc, sd, names):
        self.scale, self.nc, self.sd, self.names = scale, nc, sd, names
        self.src = []  # anchor indices of the kept detections per frame (not part of the service's JSON; tests read it)

    def detect(self, frames, conf=0.25, iou=0.7, max_det=300, precision=None):
        n = frames.shape[0]
        boxes = torch.zeros((n, max_det, 4))
        scores = torch.zeros((n, max_det))
        cls = torch.zeros((n, max_det), dtype=torch.int32)
        counts = torch.zeros((n,), dtype=torch.int32)
        for j in range(n):
            r = OY.predict(self.scale, self.nc, self.sd, np.asarray(frames[j]), conf=conf, iou=iou, max_det=max_det)
            k = len(r["src"])
            self.src.append(np.asarray(r["src"]))
            if k:
                boxes[j, :k] = torch.from_numpy(np.asarray(r["boxes"], np.float32))
                scores[j, :k] = torch.from_numpy(np.asarray(r["scores"], np.float32))
                cls[j, :k] = torch.from_numpy(np.asarray(r["cls"]).astype(np.int32))
            counts[j] = k
        return boxes, scores, cls, None, counts
