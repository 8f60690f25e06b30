"""Development aid: model-level bit reproducibility outside the bench path (SAM v1 ViT encoders, YOLOv8-pose, DINOv2)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import dino, sam, synth, weights, yolo  # noqa: E402

dev = torch.device("cuda:0")
frames = torch.from_numpy(np.stack([synth.synth_frame(100, i) for i in range(8)], 0)).to(dev)


def rep(name, fn, n=3):
    ref = fn()
    torch.cuda.synchronize()
    ok = True
    for _ in range(n):
        out = fn()
        torch.cuda.synchronize()
        ok &= all(torch.equal(a, b) for a, b in zip(out, ref))
    print(f"{name}: {'bit-reproducible' if ok else 'DIFFERS between runs'}", flush=True)


for nm, cfg in (("sam_vit_b", sam.sam_vit_b()), ("sam_vit_h", sam.sam_vit_h())):
    enc = sam.SamVitEncoder(cfg, weights.synth_state_dict(sam.vit_param_spec(cfg), 9), dev)
    rep(nm, lambda: (enc.encode(frames)["fpn"][2],))
    del enc
    torch.cuda.empty_cache()
cfg = yolo.YoloConfig("n", nc=1, kpt_shape=(17, 3))
gold = yolo.bn_stats_path("n", pose=True)
det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, gold), dev)
rep("yolov8n-pose", lambda: det.detect_pose(frames, conf=0.05))
dcfg = dino.dinov2_base()
m = dino.DinoEmbedder(dcfg, weights.synth_state_dict(dino.param_spec(dcfg), 4), dev)
rep("dinov2-base", lambda: (m.embed_frames(frames),))
