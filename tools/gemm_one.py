"""Development aid: run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  args: M N K act f32res"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

M, N, K_, act, f32 = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
a = torch.randn((M, K_), device=dev).half()
w = (torch.randn((N, K_), device=dev) * K_ ** -0.5).half()
b = torch.randn((N,), device=dev)
out = torch.randn((M, N), device=dev, dtype=torch.float32 if f32 else torch.float16)
for _ in range(5):
    K.gemm(a, w, bias=b, act=act, res=out if f32 else None, out=out)
torch.cuda.synchronize()
