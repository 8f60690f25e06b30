"""Development aid: which rows of the fused MLP differ from the unfused chain (and between runs) on identical inputs?"""
import collections
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
D = int(os.environ.get("D", "112"))
rows = int(os.environ.get("ROWS", "262144"))
x0 = torch.randn((rows, D), device=dev)
g, bb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
w1 = (torch.randn((4 * D, D), device=dev) * D ** -0.5).half()
w2 = (torch.randn((D, 4 * D), device=dev) * (4 * D) ** -0.5).half()
b1, b2 = torch.randn(4 * D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
if os.environ.get("PERIODIC"):  # every 32-hidden-unit chunk identical: a ring race (stale slot) becomes invisible
    w1 = w1[:32].repeat(4 * D // 32, 1).contiguous()
    w2 = w2[:, :32].repeat(1, 4 * D // 32).contiguous()
    b1 = b1[:32].repeat(4 * D // 32).contiguous()
xu = x0.clone()
h = K.layernorm(xu, g, bb, 1e-6)
u = K.gemm(h, w1, bias=b1, act=K.ACT_GELU)
K.gemm(u, w2, bias=b2, res=xu, out=xu)
for rep in range(4):
    o = K.ln_mlp(x0.clone(), g, bb, w1, b1, w2, b2, 1e-6)
    torch.cuda.synchronize()
    err = (o - xu).abs().amax(1)
    bad = torch.nonzero(err > 0.02).flatten().cpu().numpy()
    per = 128 if D == 112 else 256
    blocks = collections.Counter((bad // per).tolist())
    local = sorted(set((bad % per // 16).tolist()))
    cols = torch.nonzero((o - xu).abs().amax(0) > 0.02).flatten().cpu().numpy()
    if len(bad):
        m = (xu - x0)[bad].double()          # the MLP's contribution to the bad rows
        dd = (o - xu)[bad].double()
        ratio = (dd * m).sum(1) / (m * m).sum(1)
        resid = (dd - ratio[:, None] * m).norm(dim=1) / dd.norm(dim=1)
        print("   projection of the error on the row's MLP output (x 4D/32 chunks):", [round(float(v) * (4 * D // 32), 2) for v in ratio[:10]],
              "residual", [round(float(v), 2) for v in resid[:10]])
    print(f"run {rep}: {len(bad)} bad rows in {len(blocks)} blocks; 16-row groups within block: {local}; first blocks {sorted(blocks)[:8]}; "
          f"bad columns {cols[:12].tolist()}{'...' if len(cols) > 12 else ''} ({len(cols)})", flush=True)
