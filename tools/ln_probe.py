"""Development aid: LayerNorm (f32 -> f16) rate on the model's row counts, rotating over buffer sets beyond the Infinity Cache.
LMX_LN_ONE_ROW=1 selects the one-row-per-wave kernel for comparison; the two must agree bit for bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for rows, D in [(122880, 448), (30720, 896), (30150, 1024), (491520, 224)]:
    NS = max(2, min(8, -(-800_000_000 // (rows * D * 4))))
    xs = [torch.randn(rows, D, device=dev, generator=g) for _ in range(NS)]
    outs = [torch.empty(rows, D, device=dev, dtype=torch.float16) for _ in range(NS)]
    gam, bet = torch.randn(D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
    for i in range(NS):
        K.layernorm(xs[i], gam, bet, 1e-6, out=outs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(4 * NS):
        K.layernorm(xs[i % NS], gam, bet, 1e-6, out=outs[i % NS])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / (4 * NS)
    chk = int(outs[0].view(torch.int16).to(torch.int64).sum())
    print(f"rows={rows} D={D}: {us:.1f} us  {rows * D * 6 / us / 1e6:.2f} TB/s  checksum {chk}", flush=True)
