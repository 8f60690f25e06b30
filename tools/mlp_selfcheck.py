"""Development aid: run the fused MLP several times on identical inputs and count rows that differ from the first run."""
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
torch.manual_seed(0)
D, rows = 112, 1048576
x0 = torch.randn((rows, D), device=dev)
g, bb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
w1 = (torch.randn((4 * D, D), device=dev) * D ** -0.5).half()
w2 = (torch.randn((D, 4 * D), device=dev) * (4 * D) ** -0.5).half()
b1, b2 = torch.randn(4 * D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
ref = K.ln_mlp(x0.clone(), g, bb, w1, b1, w2, b2, 1e-6)
torch.cuda.synchronize()
res = []
for rep in range(5):
    o = K.ln_mlp(x0.clone(), g, bb, w1, b1, w2, b2, 1e-6)
    torch.cuda.synchronize()
    oi, ri = o.view(torch.int32), ref.view(torch.int32)
    res.append((int(((oi[:, :96] != ri[:, :96]).any(1)).sum()), int(((oi[:, 96:] != ri[:, 96:]).any(1)).sum())))
print("rows flagged 777 (b1s corrupted) in the first run:", int((ref[:, 0] == 777.0).sum()))
print(os.environ.get("LMX_DBG_LIB", "product"), "(rows whose columns 0..95 differ from the first run, rows whose columns 96..111 [xn hash in build O] differ):", res, flush=True)
