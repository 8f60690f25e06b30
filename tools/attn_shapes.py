"""Development aid: per-shape attention time table of one SAM encode + decode pass (events on the launch stream), with the
HBM floor of each launch (q, k, v read once + o written once) beside it."""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib, pipeline, synth  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
fx.serial = True
n = 16
host = np.stack([synth.synth_frame(100, i) for i in range(4)], 0)
frames = torch.from_numpy(np.concatenate([host] * (n // 4), 0)).to(dev)
fx.step(frames)
torch.cuda.synchronize()
lib = _lib.load()
raw = lib.lmx_k_attention
log = []


def traced(desc_ref, stream):
    d = desc_ref._obj
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = raw(desc_ref, stream)
    e1.record()
    log.append(((d.B, d.H, d.Tq, d.Tk, d.hd, d.mode, d.ws, d.q_stride, 1 if d.rel else 0), e0, e1))
    return rc


class P:
    def __getattr__(self, name):
        return traced if name == "lmx_k_attention" else getattr(lib, name)


_lib._lib = P()
fx.step(frames)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, a, b in log:
    c = agg.setdefault(key, [0, 0.0])
    c[0] += 1
    c[1] += a.elapsed_time(b)
tot = sum(v[1] for v in agg.values())
print(f"{len(log)} attention launches, {tot:.2f} ms for {n} frames")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    B, H, Tq, Tk, hd, mode, ws, qs, rel = k
    fl = 4.0 * B * H * Tq * Tk * hd * v[0]
    by = 2.0 * B * H * hd * (2 * Tq + 2 * Tk) * v[0]  # f16 q + o, k + v
    print(f"  B{B:6d} H{H:2d} Tq{Tq:5d} Tk{Tk:5d} hd{hd:3d} mode{mode} ws{ws:2d} qs{qs} rel{rel} x{v[0]:2d} {v[1]:8.3f} ms "
          f"{fl / v[1] / 1e9:6.0f} TF  {by / v[1] / 1e9:6.2f} TB/s")
