"""Development aid: ragged frame counts through FusedExtractor.step, and multi-stream vs single-stream equality."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import pipeline, synth  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)


def run(fr, serial):
    fx.serial = serial
    out = fx.step(fr, keep_byte_masks=True)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in out.items()}


for n in (1, 5, 17, 32):
    fr = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(n)], 0)).to(dev)
    a, b, c, d = run(fr, False), run(fr, False), run(fr, True), run(fr, True)
    msg = []
    for k in a:
        x = [float((a[k].double() - o[k].double()).abs().max()) for o in (b, c, d)]
        y = float((c[k].double() - d[k].double()).abs().max())
        if max(x) or y:
            msg.append(f"{k}: streams-vs-streams {x[0]:.3g} streams-vs-serial {x[1]:.3g}/{x[2]:.3g} serial-vs-serial {y:.3g}")
    print(n, "frames:", "; ".join(msg) if msg else "all runs bit-identical", flush=True)
