"""PCIe-inclusive rate of the fused path: frames start in pinned HOST memory (as a decoder would hand them over) and are
uploaded every step — serially before the step, and double-buffered on a copy stream under the previous step's compute."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import pipeline, synth  # noqa: E402

dev = torch.device("cuda:0")
n, steps = 64, 4
fx = pipeline.FusedExtractor(dev)
host = np.stack([synth.synth_frame(100, i) for i in range(8)], 0)
host = torch.from_numpy(np.concatenate([host] * (n // 8), 0)).pin_memory()
bufs = [torch.empty_like(host, device=dev) for _ in range(2)]
bufs[0].copy_(host)
for _ in range(2):
    fx.step(bufs[0])
torch.cuda.synchronize()

t0 = time.perf_counter()
for _ in range(steps):
    fx.step(bufs[0])
torch.cuda.synchronize()
resident = n * steps / (time.perf_counter() - t0)

t0 = time.perf_counter()
for _ in range(steps):
    bufs[0].copy_(host, non_blocking=True)
    fx.step(bufs[0])
torch.cuda.synchronize()
serial = n * steps / (time.perf_counter() - t0)

copy = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
bufs[0].copy_(host)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    cur, nxt = bufs[i & 1], bufs[(i + 1) & 1]
    copy.wait_stream(main)          # nxt was last read by step i-1, which is already enqueued on main
    with torch.cuda.stream(copy):
        nxt.copy_(host, non_blocking=True)
    fx.step(cur)
    main.wait_stream(copy)
torch.cuda.synchronize()
overlapped = n * steps / (time.perf_counter() - t0)

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
bufs[1].copy_(host, non_blocking=True)
e1.record()
torch.cuda.synchronize()
gbs = host.numel() / e0.elapsed_time(e1) / 1e6
print(f"frames resident in HBM: {resident:.0f} frames/s; uploaded serially each step: {serial:.0f}; uploaded on a copy stream "
      f"under the previous step: {overlapped:.0f}; H2D {gbs:.1f} GB/s ({host.numel() / 1e6:.0f} MB per step)")
