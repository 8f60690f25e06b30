"""Does a frame's result depend on the batch it rides in?  GEMM / conv kernels are picked by the launcher from M = n*H*W,
so a frame alone and the same frame inside a batch may go through different kernel variants.  This probe compares, bit for
bit, each kernel class on the rows of one frame alone vs the same rows inside a 3x batch, and the YOLOv8-n / -l prediction
of a frame alone vs inside a batch of 3."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import synth, yolo  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (rows, N, Kd) in [(240, 256, 128), (240, 512, 1152), (960, 128, 64), (3840, 64, 64), (240, 64, 256), (400, 96, 96)]:
    a1 = torch.randn(rows, Kd, device=dev, generator=g).half()
    w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).half()
    a3 = torch.cat([a1, torch.randn(2 * rows, Kd, device=dev, generator=g).half()], 0)
    for od in (torch.float16, torch.float32):
        o1 = K.gemm(a1, w, out_dtype=od)
        o3 = K.gemm(a3, w, out_dtype=od)[:rows]
        print(f"gemm rows={rows} N={N} K={Kd} out={od}: {'identical' if torch.equal(o1, o3) else 'DIFFERENT max ' + str(float((o1.float() - o3.float()).abs().max()))}")
for act in (K.ACT_GELU, K.ACT_SILU):  # bias + activation + LayerScale + f16 / f32 residual: the full epilogue of both kernels
    for od in (torch.float16, torch.float32):
        rows, N, Kd = 240, 256, 128
        a1 = torch.randn(rows, Kd, device=dev, generator=g).half()
        a3 = torch.cat([a1, torch.randn(2 * rows, Kd, device=dev, generator=g).half()], 0)
        w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).half()
        bias, scale = torch.randn(N, device=dev, generator=g), torch.rand(N, device=dev, generator=g) + 0.5
        r3 = torch.randn(3 * rows, N, device=dev, generator=g).to(od)
        o1 = K.gemm(a1, w, bias=bias, act=act, scale=scale, res=r3[:rows].contiguous(), out_dtype=od)
        o3 = K.gemm(a3, w, bias=bias, act=act, scale=scale, res=r3, out_dtype=od)[:rows]
        print(f"gemm epilogue act={act} out={od}: {'identical' if torch.equal(o1, o3) else 'DIFFERENT max ' + str(float((o1.float() - o3.float()).abs().max()))}")
for (h, w_, cin, cout) in [(12, 20, 256, 256), (24, 40, 128, 128), (48, 80, 64, 64), (12, 20, 512, 512)]:
    x1 = torch.randn(1, h, w_, cin, device=dev, generator=g).half()
    x3 = torch.cat([x1, torch.randn(2, h, w_, cin, device=dev, generator=g).half()], 0)
    wt = (torch.randn(cout, 9 * cin, device=dev, generator=g) * (9 * cin) ** -0.5).half()
    b = torch.zeros(cout, device=dev)
    o1 = K.conv3x3(x1, wt, b, act=K.ACT_SILU)
    o3 = K.conv3x3(x3, wt, b, act=K.ACT_SILU)[:1]
    print(f"conv3x3 {h}x{w_} cin={cin} cout={cout}: {'identical' if torch.equal(o1, o3) else 'DIFFERENT max ' + str(float((o1.float() - o3.float()).abs().max()))}")
for scale in ("n", "l"):
    cfg = yolo.YoloConfig(scale)
    det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path(scale)), dev)
    fr = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + 3 * i) for i in range(3)], 0)).to(dev)
    img, _ = det.preprocess(fr)
    p3 = det.forward_letterboxed(img)
    p1 = det.forward_letterboxed(img[:1])
    d = (p3[0] - p1[0]).abs()
    print(f"yolov8{scale} frame alone vs in a batch of 3: {'identical' if torch.equal(p3[0], p1[0]) else 'DIFFERENT'} max box {float(d[:, :4].max()):.4f} max score {float(d[:, 4:].max()):.2e}")
