"""How far the fp32 CPU oracle of YOLOv8 sits from an f64 evaluation of the same network (same f32-folded weights): the
resolution below which the order of two scores is not a property of the reference but of its summation order.
CPU only.  Usage: python tools/fp32_noise_probe.py [scale] (cfg#2 golden frame 31 for l, the (3,40) clip frame for n)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import synth, yolo  # noqa: E402
from oracle import yolo as OY  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "l"
cfg = yolo.YoloConfig(scale)
sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path(scale))
frame = synth.cfg2_frames()[31] if scale == "l" else synth.synth_frame(3, 40)
lb = OY.letterbox(frame)
x = torch.from_numpy(np.ascontiguousarray(lb[:, :, ::-1].transpose(2, 0, 1))).float()[None] / 255
with torch.no_grad():
    p32 = OY.model_forward(scale, cfg.nc, sd, x)[0].transpose(0, 1).numpy()


def fc(sd_, name, xx, k, s, act=True):  # the oracle's fused conv with the SAME f32-folded weights, evaluated in f64
    w = OY._t(sd_, name + ".conv.weight")
    g, b = OY._t(sd_, name + ".bn.weight"), OY._t(sd_, name + ".bn.bias")
    mu, var = OY._t(sd_, name + ".bn.running_mean"), OY._t(sd_, name + ".bn.running_var")
    sc = g / torch.sqrt(var + 1e-3)
    y = _conv(xx, (w * sc.view(-1, 1, 1, 1)).double(), (b - mu * sc).double(), stride=s, padding=k // 2)
    return y / (1 + torch.exp(-y)) if act else y


_conv = F.conv2d
OY._fused_conv = fc
F.conv2d = lambda a, w, b=None, **kw: _conv(a.double(), w.double(), None if b is None else b.double(), **kw)
with torch.no_grad():
    p64 = OY.model_forward(scale, cfg.nc, sd, x.double())[0].transpose(0, 1).numpy()
s32, s64 = p32[:, 4:].max(1), p64[:, 4:].max(1)
d = np.abs(s32 - s64)
print(f"yolov8{scale}: fp32 oracle vs f64 evaluation: best-class score max |diff| {d.max():.3e}, mean {d.mean():.3e}, "
      f"99.9th percentile {np.quantile(d, 0.999):.3e}; boxes max {np.abs(p32[:, :4] - p64[:, :4]).max():.3e} px")
if scale == "l":
    for a in (6576, 5884):
        print(f"  anchor {a}: fp32 {s32[a]:.9f}  f64 {s64[a]:.9f}")
