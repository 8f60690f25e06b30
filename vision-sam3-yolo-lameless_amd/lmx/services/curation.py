"""clip-curation's per-frame YOLO consumer on liblmx: `detect_cow_in_frame`
(services/clip-curation/app/main.py:103-131) and the per-frame records of `track_cow_through_video` (:154-165), batched
over the frames of a clip.  The detector runs on every frame of 30 s uploads — the densest YOLO consumer of the product —
so the frames go through YoloDetector.detect in batches and only the [n, max_det] box / score / class tables (7 kB per frame)
come back to the host, where the reference's selection rule is applied unchanged."""
import numpy as np

COCO_COW = 19  # "Class 19 is cow in COCO" (main.py:114)


def best_detection(boxes, scores, cls, count, frame_h, frame_w):
    """main.py:106-131 on one frame's NMS output: the largest box that is a cow or covers more than 10 % of the frame."""
    best, best_area = None, 0
    frame_area = frame_h * frame_w
    for j in range(int(count)):
        x1, y1, x2, y2 = (np.float32(v) for v in boxes[j])
        area = (x2 - x1) * (y2 - y1)
        if (int(cls[j]) == COCO_COW or area > frame_area * 0.1) and area > best_area:
            best_area = area
            best = {"bbox": [float(x1), float(y1), float(x2), float(y2)], "confidence": float(scores[j]),
                    "centroid": ((x1 + x2) / 2, (y1 + y2) / 2), "area": area}
    return best


class CowTracker:
    def __init__(self, detector, conf=0.3, batch=64):
        self.detector, self.conf, self.batch = detector, conf, batch

    def track(self, frames, fps, first_frame=0):
        """frames: u8 BGR [n,h,w,3] on the detector's device -> the `detections` list of track_cow_through_video:
        [{"frame", "time", "detection": best_detection | None}] (main.py:154-165)."""
        n, h, w, _ = frames.shape
        out = []
        for i in range(0, n, self.batch):
            boxes, scores, cls, _, counts = (t.cpu().numpy() for t in self.detector.detect(frames[i:i + self.batch], conf=self.conf))
            for b in range(boxes.shape[0]):
                idx = first_frame + i + b
                out.append({"frame": idx, "time": idx / fps if fps > 0 else 0,
                            "detection": best_detection(boxes[b], scores[b], cls[b], counts[b], h, w)})
        return out
