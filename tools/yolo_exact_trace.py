"""Per-launch table of YOLOv8-l's exact plan on the reference schedule's 10 frames (event-timed on one stream)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import synth, yolo  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "exact"
dev = torch.device("cuda:0")
cfg = yolo.YoloConfig("l")
det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("l")), dev)
d = torch.from_numpy(np.stack([synth.synth_frame(5, i) for i in range(10)], 0)).to(dev)
for _ in range(3):
    det.detect(d, conf=0.5, precision=prec)
torch.cuda.synchronize()
K.start_launch_trace()
for _ in range(5):
    det.detect(d, conf=0.5, precision=prec)
tr, shapes = K.stop_launch_trace(by_shape=True)
tot = sum(r["seconds"] for r in shapes.values())
print(f"plan {prec}: {tot / 5 * 1e3:.2f} ms per detect() of 10 frames (event-timed), {sum(r['launches'] for r in shapes.values()) // 5} launches")
for (cls, key), r in sorted(shapes.items(), key=lambda kv: -kv[1]["seconds"])[:28]:
    print(f"{100 * r['seconds'] / tot:5.1f}%  {r['launches'] // 5:3d}x {r['seconds'] / r['launches'] * 1e6:7.1f} us  {r['flops'] / max(r['seconds'], 1e-12) / 1e12:6.0f} TF  [{cls}] {key}")
