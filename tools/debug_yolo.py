"""Development aid: layer-by-layer comparison of the HIP YOLO path with the fp32 oracle on one frame."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import synth, yolo  # noqa: E402
import oracle.yolo as OY  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "n"
cfg = yolo.YoloConfig(scale)
sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path(scale))
dev = torch.device("cuda:0")
det = yolo.YoloDetector(cfg, sd, dev)
f = synth.synth_frame(3, 40)
# oracle features per fused conv
feats = {}
orig = OY._fused_conv


def spy(sd_, name, xx, k, s, act=True):
    y = orig(sd_, name, xx, k, s, act)
    feats[name] = y
    return y


OY._fused_conv = spy
lb = OY.letterbox(f)
x = torch.from_numpy(np.ascontiguousarray(lb[:, :, ::-1].transpose(2, 0, 1))).float()[None] / 255
with torch.no_grad():
    ref_pred = OY.model_forward(scale, 80, sd, x)[0].transpose(0, 1)
OY._fused_conv = orig

# GPU: capture outputs of conv kernels by wrapping K.conv3x3 / conv1x1 / stem_conv
cap = []
o3, o1, os_ = K.conv3x3, K.conv1x1, K.stem_conv


def w3(xx, w, bias=None, act=K.ACT_SILU, stride=1, res=None, out=None):
    y = o3(xx, w, bias=bias, act=act, stride=stride, res=res, out=out)
    cap.append(("c3", y, res is not None))
    return y


def w1(xx, w, bias=None, act=K.ACT_SILU, res=None, out=None, out_dtype=torch.float16):
    y = o1(xx, w, bias=bias, act=act, res=res, out=out, out_dtype=out_dtype)
    cap.append(("c1", y, False))
    return y


def ws(img, w, b, out=None):
    y = os_(img, w, b, out)
    cap.append(("stem", y, False))
    return y


K.conv3x3, K.conv1x1, K.stem_conv = w3, w1, ws
img, geo = det.preprocess(torch.from_numpy(f[None]).to(dev))
assert np.array_equal(img[0].cpu().numpy(), lb[:, :, ::-1])
pred = det.forward_letterboxed(img)
torch.cuda.synchronize()
# the oracle's conv order equals the launch order except residual adds (oracle stores pre-add t)
names = list(feats.keys())
print(len(names), len(cap))
for (kind, y, has_res), name in zip(cap, names):
    r = feats[name][0].permute(1, 2, 0)
    g = y[0].float().cpu()
    if g.shape != r.shape:
        print(name, "shape mismatch", tuple(g.shape), tuple(r.shape))
        continue
    if has_res:
        print(f"{name:28s} (residual fused, skipped)")
        continue
    err = (g - r).abs()
    print(f"{name:28s} {kind} rms_ref {float(r.pow(2).mean().sqrt()):.3f} max_err {float(err.max()):.4f} mean_err {float(err.mean()):.5f}")
p = pred[0].cpu()
print("pred box err", float((p[:, :4] - ref_pred[:, :4]).abs().max()), "cls err", float((p[:, 4:] - ref_pred[:, 4:]).abs().max()))
e = (p[:, :4] - ref_pred[:, :4]).abs().max(1).values
print("anchors with box err > 1px:", int((e > 1).sum()), "of", len(e), " > 0.1:", int((e > 0.1).sum()))
