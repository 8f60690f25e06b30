"""Development aid: per-tile fixed cost vs per-k-step cost of the GEMM (M=65536, N=1792, K swept)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib  # noqa: E402
if os.environ.get("LMX_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LMX_DBG_LIB"])
from lmx import kernels as K  # noqa: E402
from perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
M = 65536
for (N, act, f32, res) in [(1792, 0, 0, 0)] + ([] if os.environ.get('LMX_DBG_LIB') else [(1792, 2, 0, 0), (1792, 0, 1, 1), (448, 0, 1, 1)]):
    for K_ in (32, 64, 128, 256, 448, 896, 1792, 3584):
        a = torch.randn((M, K_), device=dev).half()
        w = (torch.randn((N, K_), device=dev) * K_ ** -0.5).half()
        b = torch.randn((N,), device=dev)
        out = torch.zeros((M, N), device=dev, dtype=torch.float32 if f32 else torch.float16)
        ms = timeit(lambda: K.gemm(a, w, bias=b, act=act, res=out if res else None, out=out), iters=10)
        tiles = (M // 256) * ((N + 127) // 128)
        print(f"N={N} act{act} {'f32' if f32 else 'f16'} res{res} K={K_:5d}: {ms * 1e3:8.1f} us  {2 * M * N * K_ / ms / 1e9:7.1f} TF  "
              f"per-tile-slot {ms * 1e3 * 512 / tiles:6.2f} us", flush=True)
