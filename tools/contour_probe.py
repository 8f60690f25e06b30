"""Development aid: time lmx_k_contour_features on the masks the bench produces (synthetic-weight SAM masks are noisy) and
on smooth blobs, 16 frames of 1080p each; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import pipeline, synth  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
frames = torch.from_numpy(synth.synth_clip(100, 16)).to(dev)
out = fx.step(frames, keep_byte_masks=True)
torch.cuda.synchronize()
sam_masks = out["mask"].contiguous()
yy, xx = np.mgrid[0:1080, 0:1920]
blob = ((xx - 900) ** 2 / 4 + (yy - 500) ** 2 <= 300 ** 2).astype(np.uint8)
blobs = torch.from_numpy(np.stack([blob] * 16, 0)).to(dev)
for name, m in (("SAM masks (synthetic weights)", sam_masks), ("smooth blobs", blobs)):
    K.contour_features(m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        c = K.contour_features(m)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 5:.3f} ms per 16 frames; coverage {float((m != 0).float().mean()):.3f}; "
          f"external contours per frame {c[:, 7].tolist()}; unit/diag steps frame 0: {c[0, 1:3].tolist()}", flush=True)
