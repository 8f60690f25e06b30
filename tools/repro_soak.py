"""Development aid: N multi-stream FusedExtractor steps (current LMX_MAX_STREAMS / LMX_STREAM_LAYOUT) against the one-stream
result of the same frames, every output field compared bit for bit.   python tools/repro_soak.py [frames] [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(n)], 0)).to(dev)
fx.serial = True
ref = {k: v.clone() for k, v in fx.step(frames, keep_byte_masks=True).items()}
torch.cuda.synchronize()
fx.serial = False
dirty = 0
for rep in range(steps):
    out = fx.step(frames, keep_byte_masks=True)
    torch.cuda.synchronize()
    diff = {k: int((out[k] != ref[k]).sum()) for k in ref if not torch.equal(out[k], ref[k])}
    if diff:
        dirty += 1
        print(f"step {rep}: differs from the one-stream result: {diff}", flush=True)
print(f"layout={fx.stream_layout} max_streams={fx.max_streams} frames={n}: {steps - dirty} of {steps} steps bit-identical to the "
      f"one-stream result", flush=True)
