"""CPU stand-ins for the lmx.kernels wrappers the YOLO launch plans call — TEST INFRASTRUCTURE.

They let the `-m "not gpu"` suite walk lmx.yolo's launch plans (buffer slicing, channel-group layouts of the exact plan's
x3 format, weight packing) on the CPU: each stand-in does what the C-ABI entry point documents (include/lmx.h), in float64
where the device accumulates in f32.  Nothing in the product imports this file; the GPU tests never use it."""
import numpy as np
import torch
import torch.nn.functional as F


def _act(t, act):
    if act == 1:
        return t / (1 + torch.exp(-t))
    if act == 3:
        return t.clamp_min(0)
    return t


def conv3x3(x, w, bias=None, act=1, stride=1, res=None, out=None, scale=None, out_dtype=torch.float16, split_k=1):
    n, H, W, cin = x.shape
    cout = w.shape[0]
    wk = w.double().view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), wk, None if bias is None else bias.double(), stride=stride, padding=1)
    y = _act(y, act)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if out is None:
        out = torch.empty(y.shape, dtype=out_dtype)
    if res is not None:
        y = y.to(out.dtype).double() + res.double()
    out.copy_(y.to(out.dtype))
    return out


def conv1x1(x, w, bias=None, act=1, res=None, out=None, out_dtype=torch.float16, scale=None):
    y = x.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    y = _act(y, act)
    if scale is not None:
        y = y * scale.double()
    if out is None:
        out = torch.empty(y.shape, dtype=out_dtype)
    out.copy_(y.to(out.dtype))
    return out


def stem_conv(img, w, bias, out=None):
    x = img.permute(0, 3, 1, 2).float() / 255
    y = F.silu(F.conv2d(x, w.permute(3, 2, 0, 1).contiguous(), bias, stride=2, padding=1)).permute(0, 2, 3, 1)
    return y.half()


def _split(v):
    hi = v.float().half()
    lo = ((v.float() - hi.float()) * 2048).half()
    return hi, lo


def _join(t3, g):
    n, H, W, c3 = t3.shape
    t = t3.view(n, H, W, c3 // (3 * g), 3, g)
    return (t[..., 0, :].double() + t[..., 1, :].double() / 2048).reshape(n, H, W, c3 // 3)


def _write3(v, out3, g):
    n, H, W, N = v.shape
    hi, lo = _split(v)
    hi, lo = hi.view(n, H, W, N // g, g), lo.view(n, H, W, N // g, g)
    t = torch.stack((hi, lo, hi), 4).reshape(n, H, W, 3 * N)
    out3.copy_(t)
    return out3


def stem_conv_x3(img, w, bias):
    x = img.permute(0, 3, 1, 2).float() / 255
    y = F.silu(F.conv2d(x, w.permute(3, 2, 0, 1).contiguous(), bias, stride=2, padding=1)).permute(0, 2, 3, 1)
    out = torch.empty(y.shape[:3] + (3 * y.shape[3],), dtype=torch.float16)
    return _write3(y, out, y.shape[3])


def split3(x, act, out3, g=None, res3=None):
    g = g or x.shape[3]
    v = _act(x.float(), act)
    if res3 is not None:
        v = (v.double() + _join(res3, g)).float()
    return _write3(v, out3, g)


def maxpool5(x, out):
    out.copy_(F.max_pool2d(x.float().permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1).half())
    return out


def maxpool5_x3(x3, out3):
    g = x3.shape[3] // 3
    v = _join(x3, g)
    m = F.max_pool2d(v.permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1)
    return _write3(m.float(), out3, g)


def upsample2(x, out):
    out.copy_(x.repeat_interleave(2, 1).repeat_interleave(2, 2))
    return out


def detect_decode(head, pred, nc, stride, a_off):
    n, H, W, _ = head.shape
    box = head[..., :64].reshape(n, H * W, 4, 16).softmax(-1)
    dist = (box * torch.arange(16.0)).sum(-1)
    gy, gx = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    a = torch.stack((gx, gy), -1).view(-1, 2)
    x1y1, x2y2 = a - dist[..., :2], a + dist[..., 2:]
    pred[:, a_off:a_off + H * W] = torch.cat(((x1y1 + x2y2) / 2 * stride, (x2y2 - x1y1) * stride,
                                              head[..., 64:64 + nc].reshape(n, H * W, nc).sigmoid()), -1)
    return pred


def install(monkeypatch):
    from lmx import kernels as K

    monkeypatch.setattr(K, "split_k_for", lambda *a: 1)

    for name in ("conv3x3", "conv1x1", "stem_conv", "stem_conv_x3", "split3", "maxpool5", "maxpool5_x3", "upsample2", "detect_decode"):
        monkeypatch.setattr(K, name, globals()[name])
    from lmx import yolo

    monkeypatch.setattr(yolo._PlanF16, "pool5", staticmethod(maxpool5))
    monkeypatch.setattr(yolo._PlanF16, "up2", staticmethod(upsample2))
    monkeypatch.setattr(yolo._PlanExact, "pool5", staticmethod(maxpool5_x3))
    monkeypatch.setattr(yolo._PlanExact, "up2", staticmethod(upsample2))
