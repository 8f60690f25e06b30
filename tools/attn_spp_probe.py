"""A/B of the whole-sequence attention kernels (csrc/attn.hip): attn_sp_kernel (LMX_ATTN_NO_SPP=1) vs the persistent
double-buffered attn_spp_kernel, at the bench's shapes: Hiera-B+ stage 3 windows (30 frames: B = 750, H = 8, 196 tokens, hd 56),
its q-pooled form (Tq = 49) and DINOv3's 201 tokens (B = 150, H = 16, hd 64).  Prints time per launch (cold: three tensor sets
rotated) and a digest of the output so that two runs can be compared bit for bit."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
tag = "attn_sp (LMX_ATTN_NO_SPP)" if os.environ.get("LMX_ATTN_NO_SPP") else "attn_spp (persistent)"


def bench(name, make, n=12, sets=3):
    data = [make() for _ in range(sets)]
    for i in range(3):
        data[i % sets][0]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        data[i % sets][0]()
    e1.record()
    torch.cuda.synchronize()
    dig = hashlib.sha1(data[0][1].cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{tag:28s} {name:44s} {e0.elapsed_time(e1) * 1000 / n:8.1f} us   out sha1 {dig}  finite {bool(torch.isfinite(data[0][1].float()).all())}", flush=True)


def window(frames, D, H, hd, G, ws, qs=1):
    nW = (-(-G // ws)) ** 2
    rows = frames * G * G

    def make():
        qkv = torch.randn(rows, 3 * D, device=dev, generator=g).half()
        padkv = torch.randn(3 * D, device=dev, generator=g).half()
        if qs == 2:
            q = torch.randn(rows // 4, D, device=dev, generator=g).half()
            out = torch.empty(rows // 4, D, device=dev, dtype=torch.float16)
        else:
            q, out = qkv[:, :D], torch.empty(rows, D, device=dev, dtype=torch.float16)
        wq = ws // qs
        return (lambda: K.attention(q, qkv[:, D:2 * D], qkv[:, 2 * D:], out, frames * nW, H, wq * wq, ws * ws, hd, hd ** -0.5,
                                    window=dict(Gh=G, Gw=G, ws=ws, q_stride=qs), pad_k=padkv[D:2 * D], pad_v=padkv[2 * D:])), out

    return make


def flat(B, H, T, hd):
    D = H * hd

    def make():
        qkv = torch.randn(B * T, 3 * D, device=dev, generator=g).half()
        out = torch.empty(B * T, D, device=dev, dtype=torch.float16)
        return (lambda: K.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, B, H, T, T, hd, hd ** -0.5)), out

    return make


bench("hiera stage 3 windows 14x14, 30 frames", window(30, 448, 8, 56, 64, 14))
bench("hiera stage 3 windows 14x14, 10 frames", window(10, 448, 8, 56, 64, 14))
bench("dinov3 201 tokens, 150 frames, hd 64", flat(150, 16, 201, 64))
bench("dinov3 201 tokens, 5 frames, hd 64", flat(5, 16, 201, 64))
