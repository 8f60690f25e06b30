"""The model half of tleap-pipeline's pose estimation on liblmx: what `self.model(frame, verbose=False, conf=0.3)` and the
`result.boxes[j]` / `result.keypoints[j].data[0]` unpacking produce (services/tleap-pipeline/app/main.py:142-171) for a
batch of frames.  SURVEY.md §8f rank 2: same trunk and kernels as the detector, plus the Pose head (`lmx.yolo`, cv4 branch +
`lmx_k_pose_gather`).  The bounding-box heuristics tleap blends in afterwards (main.py:173-200) are host-side product logic
and stay where they are: this returns the `model_keypoints` dict they start from."""
import numpy as np

# services/tleap-pipeline/app/main.py:43-64 — the 20 cow keypoints of the project's trained pose model
KEYPOINT_NAMES = [
    "left_ear_base", "neck", "withers", "mid_back", "right_hind_hip", "right_hind_mid_leg", "right_hind_fetlock",
    "left_hind_shoulder", "left_hind_mid_leg", "left_hind_fetlock", "right_front_shoulder", "right_front_mid_leg",
    "right_front_lower_leg", "left_front_shoulder", "left_front_mid_leg", "left_front_lower_leg", "right_front_hoof",
    "left_front_hoof", "right_hind_hoof", "left_hind_hoof",
]


class PoseEstimator:
    def __init__(self, detector, keypoint_names=None, conf=0.3):
        """detector: lmx.yolo.YoloDetector built with YoloConfig(kpt_shape=(K, 2|3))."""
        if detector.cfg.kpt_shape is None:
            raise ValueError("PoseEstimator needs a pose model (YoloConfig.kpt_shape)")
        self.detector, self.conf = detector, conf
        self.names = list(keypoint_names if keypoint_names is not None else KEYPOINT_NAMES)

    def detect_with_trained_model(self, frames):
        """frames: u8 BGR [n,h,w,3] on the detector's device -> per frame, a list of
        {"bbox": [x1,y1,x2,y2], "confidence": float, "model_keypoints": {name: {"name","x","y","confidence"}}}
        (main.py:152-171; keypoints beyond the name list are called kp_{i}, a 2-d keypoint has confidence 1.0)."""
        boxes, scores, cls, src, counts, kpts = (t.cpu().numpy() for t in self.detector.detect_pose(frames, conf=self.conf))
        out = []
        for b in range(boxes.shape[0]):
            dets = []
            for j in range(int(counts[b])):
                mk = {}
                for i, kp in enumerate(kpts[b, j]):
                    name = self.names[i] if i < len(self.names) else f"kp_{i}"
                    mk[name] = {"name": name, "x": float(kp[0]), "y": float(kp[1]),
                                "confidence": float(kp[2]) if len(kp) > 2 else 1.0}
                dets.append({"bbox": np.asarray(boxes[b, j], np.float64).tolist(), "confidence": float(scores[b, j]),
                             "model_keypoints": mk})
            out.append(dets)
        return out
