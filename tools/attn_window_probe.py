"""Development aid: is the 14 x 14-window attention of Hiera stage 3 (B = 25 windows x frames, H = 8, 196 tokens, hd = 56) bound
by the latency of its cold K / V / Q reads?  Times it per item on qkv tensors that fit the Infinity Cache (hot, looped) and on
the bench's 30-frame tensor rotated over several copies (cold)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
D, H, hd, G, ws = 448, 8, 56, 64, 14
nW = (-(-G // ws)) ** 2
for frames, sets in [(4, 1), (8, 1), (30, 1), (30, 3)]:
    rows = frames * G * G
    qkvs = [torch.randn(rows, 3 * D, device=dev, generator=g).half() for _ in range(sets)]
    outs = [torch.empty(rows, D, device=dev, dtype=torch.float16) for _ in range(sets)]
    padkv = torch.randn(3 * D, device=dev, generator=g).half()

    def run(i):
        q = qkvs[i % sets]
        K.attention(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], outs[i % sets], frames * nW, H, ws * ws, ws * ws, hd, hd ** -0.5,
                    window=dict(Gh=G, Gw=G, ws=ws, q_stride=1), pad_k=padkv[D:2 * D], pad_v=padkv[2 * D:])

    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 12
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    items = frames * nW * H
    by = rows * D * 2 * 4
    print(f"{frames:2d} frames x {sets} set(s) (qkv {rows * 3 * D * 2 / 1e6:.0f} MB each): {us:7.1f} us  {us * 1000 / items:6.1f} ns per (window, head)  "
          f"{by / us / 1e6:.2f} TB/s", flush=True)
