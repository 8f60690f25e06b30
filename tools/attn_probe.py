"""Development aid: attention kernel timing at the three shapes that matter (DINO, Hiera global, Hiera window 14)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib  # noqa: E402
if os.environ.get("LMX_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LMX_DBG_LIB"])
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(name, B, H, T, hd, window=None, n_img=None):
    D = H * hd
    rows = B * T if window is None else n_img * window["Gh"] * window["Gw"]
    qkv = torch.randn((rows, 3 * D), device=dev).half()
    o = torch.empty((rows, D), device=dev, dtype=torch.float16)
    pad = torch.randn((3 * D,), device=dev).half()
    ms = timeit(lambda: K.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, T, T, hd, hd ** -0.5, window=window,
                                    pad_k=pad[D:2 * D], pad_v=pad[2 * D:]))
    print(f"{name}: {ms:.3f} ms  {4.0 * B * H * T * T * hd / ms / 1e9:.0f} TFLOP/s (useful)", flush=True)


run("dino   B256 H16 T201 hd64", 256, 16, 201, 64)
run("global B16  H8  T4096 hd56", 16, 8, 4096, 56)
run("win14  16img H8 T196 hd56", 16 * 25, 8, 196, 56, window=dict(Gh=64, Gw=64, ws=14, q_stride=1), n_img=16)
run("win8   16img H2 T64 hd56", 16 * 1024, 2, 64, 56, window=dict(Gh=256, Gw=256, ws=8, q_stride=1), n_img=16)
