"""Development aid: the Pillow-exact preprocessing of the path on 16 x 1080p frames (SAM: bilinear 1920 -> 1024 + im2col;
DINO: bicubic 1920 -> 455 + patchify)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib  # noqa: E402
if os.environ.get("LMX_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LMX_DBG_LIB"])
from lmx import pipeline, synth  # noqa: E402
from perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
host = np.stack([synth.synth_frame(100, i) for i in range(4)], 0)
frames = torch.from_numpy(np.concatenate([host] * 4, 0)).to(dev)
for name, fn in (("sam preprocess", lambda: fx.sam.preprocess(frames)), ("dino preprocess", lambda: fx.dino.preprocess(frames))):
    ms = timeit(fn, iters=10)
    print(f"{name} 16 x 1080p: {ms:.3f} ms  ({frames.numel() / ms / 1e6:.0f} GB/s of input)", flush=True)
