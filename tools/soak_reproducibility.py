"""Soak: the fused step, repeated on the multi-stream schedule, must give the same bits every time (every output field,
including the 1080p masks).  usage: python tools/soak_reproducibility.py [steps=40] [frames=60]
(the round-1 defect — DESIGN.md section 6 — showed as run-to-run differences of exactly this kind; tests/test_gpu_services.py
holds the six-run version of this check, tests/test_gpu_defect.py the victims beside the aggressor)"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import pipeline, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
frames = torch.from_numpy(synth.synth_clip(7, n)).to(dev)


def digest(out):
    h = hashlib.sha256()
    for k in sorted(out):
        h.update(k.encode())
        h.update(out[k].contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


seen = {}
for i in range(steps):
    fx.serial = (i == 0)  # the first pass on one stream is the reference
    out = fx.step(frames, sam_chunk=30, keep_byte_masks=True)
    torch.cuda.synchronize()
    d = digest(out)
    seen[d] = seen.get(d, 0) + 1
    if i % 10 == 0:
        print(f"step {i}: {len(seen)} distinct digest(s) so far", flush=True)
print(f"{steps} steps of {n} frames ({fx.max_streams} streams; step 0 on one stream): {len(seen)} distinct output digest(s) "
      f"{'— bit-reproducible' if len(seen) == 1 else '— NOT REPRODUCIBLE: ' + str(sorted(seen.values()))}")
sys.exit(0 if len(seen) == 1 else 1)
