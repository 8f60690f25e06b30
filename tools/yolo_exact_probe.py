"""Exact vs f16 plan of YOLOv8-l on the reference schedule's 10 frames per clip (1080p): wall time per call on one stream,
and the deviation of both plans from the committed fp32 golden sample.  Usage: python tools/yolo_exact_probe.py [scale]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import synth, yolo  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "l"
dev = torch.device("cuda:0")
cfg = yolo.YoloConfig(scale)
det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path(scale)), dev)
gold = np.load(os.path.join(ROOT, "tests", "golden", f"yolov8{scale}_det_w7.npz"))
fr = np.stack([synth.synth_frame(3, 40), synth.synth_frame(2, 50)] + [synth.synth_frame(5, i) for i in range(8)], 0)
d = torch.from_numpy(fr).to(dev)
for prec in ("f16", "exact"):
    img, _ = det.preprocess(d)
    pred = det.forward_letterboxed(img, prec)
    torch.cuda.synchronize()
    p = pred.cpu().numpy()
    for j in range(2):
        ref = gold[f"f{j}_pred_sample"]
        got = p[j][::97]
        print(f"{prec:5s} frame {j}: max |score - fp32| {np.abs(got[:, 4:] - ref[:, 4:]).max():.3e}, max |box - fp32| {np.abs(got[:, :4] - ref[:, :4]).max():.3e} px")
    for _ in range(3):
        det.detect(d, conf=0.5, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        det.detect(d, conf=0.5, precision=prec)
    torch.cuda.synchronize()
    print(f"{prec:5s}: {(time.perf_counter() - t0) * 100:.2f} ms per detect() of {d.shape[0]} frames")
