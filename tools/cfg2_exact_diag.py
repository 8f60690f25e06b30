"""Where the exact plan's keep-set differs from the fp32 golden on BASELINE cfg#2's golden frames: prints the differing
positions with the fp32 scores around them."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import synth, yolo  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "yolov8l_cfg2_w7.npz"))
cfg = yolo.YoloConfig("l")
det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("l")), dev)
fr = synth.cfg2_frames()[[int(i) for i in g["frame_ids"]]]
d = torch.from_numpy(fr).to(dev)
img, _ = det.preprocess(d)
pred = det.forward_letterboxed(img).cpu().numpy()
for conf in (0.25, 0.5):
    b, s, c, src, cnt = (t.cpu().numpy() for t in det.detect(d, conf=conf))
    for j in range(2):
        gs = g[f"f{j}_c{int(conf * 100)}_src"]
        k = int(cnt[j])
        dv = src[j, :k]
        if k == len(gs) and np.array_equal(dv, gs):
            print(f"frame {j} conf {conf}: equal ({k})")
            continue
        print(f"frame {j} conf {conf}: device {k} vs golden {len(gs)}; set difference dev-gold {sorted(set(dv) - set(gs))} gold-dev {sorted(set(gs) - set(dv))}")
        sc32 = g[f"f{j}_score"]
        scd = pred[j][:, 4:].max(1)
        for pos in np.nonzero(dv[:min(k, len(gs))] != gs[:min(k, len(gs))])[0][:12]:
            a, bb = int(dv[pos]), int(gs[pos])
            print(f"  pos {pos}: device anchor {a} (fp32 score {sc32[a]:.9f}, device {scd[a]:.9f})  golden anchor {bb} (fp32 {sc32[bb]:.9f}, device {scd[bb]:.9f})")
