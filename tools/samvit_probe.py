"""Development aid: SAM v1 ImageEncoderViT b / l / h at full size (1024 x 1024) from 1080p frames: time and sanity."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import sam, synth, weights  # noqa: E402
from perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
n = 8
frames = torch.from_numpy(np.stack([synth.synth_frame(100, i) for i in range(n)], 0)).to(dev)
for name, cfg in (("vit_b", sam.sam_vit_b()), ("vit_l", sam.sam_vit_l()), ("vit_h", sam.sam_vit_h())):
    t = time.time()
    enc = sam.SamVitEncoder(cfg, weights.synth_state_dict(sam.vit_param_spec(cfg), 9), dev)
    out = enc.encode(frames)["fpn"][2]
    torch.cuda.synchronize()
    ms = timeit(lambda: enc.encode(frames), iters=3, warm=1)
    print(f"{name}: {ms:.1f} ms for {n} frames ({n / ms * 1e3:.0f} frames/s), embedding {tuple(out.shape)} finite={bool(torch.isfinite(out.float()).all())} "
          f"rms={float(out.float().pow(2).mean().sqrt()):.3f}  (build {time.time() - t:.0f}s)", flush=True)
    del enc
    torch.cuda.empty_cache()
