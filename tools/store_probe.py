"""Development aid: which part of a short-K, store-dominated GEMM is slow?  (1M rows, K=112, f16 out)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
M = 1048576
for (N, K_, act, f32, res) in [(448, 112, 0, 0, 0), (448, 112, 2, 0, 0), (448, 112, 1, 0, 0), (512, 128, 0, 0, 0), (384, 128, 0, 0, 0),
                               (336, 112, 0, 0, 0), (256, 128, 0, 0, 0), (128, 128, 0, 0, 0), (448, 112, 0, 1, 0), (448, 112, 0, 1, 1),
                               (128, 512, 0, 0, 0), (128, 512, 0, 1, 0), (128, 512, 0, 1, 1), (448, 32, 0, 0, 0)]:
    a = torch.randn((M, K_), device=dev).half()
    w = (torch.randn((N, K_), device=dev) * K_ ** -0.5).half()
    b = torch.randn((N,), device=dev)
    out = torch.zeros((M, N), device=dev, dtype=torch.float32 if f32 else torch.float16)
    ms = timeit(lambda: K.gemm(a, w, bias=b, act=act, res=out if res else None, out=out), iters=10)
    nb = M * K_ * 2 + M * N * out.element_size() * (2 if res else 1)
    print(f"N={N} K={K_} act{act} {'f32' if f32 else 'f16'} res{res}: {ms:.3f} ms  {nb / ms / 1e9:.2f} TB/s  "
          f"store {M * N * out.element_size() / ms / 1e9:.2f} TB/s", flush=True)
x = torch.empty((M, 448), device=dev, dtype=torch.float16)
y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x), iters=10)
print(f"torch copy 940 MB: {ms:.3f} ms {2 * x.numel() * 2 / ms / 1e9:.2f} TB/s")
ms = timeit(lambda: y.fill_(1.0), iters=10)
print(f"torch fill 940 MB: {ms:.3f} ms {x.numel() * 2 / ms / 1e9:.2f} TB/s")
