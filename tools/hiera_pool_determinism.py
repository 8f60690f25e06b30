"""Development aid: is lmx_k_hiera_attn_pool bit-reproducible when other kernels share the GPU (a second stream running GEMMs)?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import sam  # noqa: E402

dev = torch.device("cuda:0")
Din, D, heads = 112, 224, 4
n, G = 4, 256
g = torch.Generator().manual_seed(7)
h = torch.randn((n * G * G, Din), generator=g).half().to(dev)
wsc = (torch.randn((D, Din), generator=g) * Din ** -0.5).half().float()
bsc = torch.randn((D,), generator=g) * 0.2
wqkv = (torch.randn((3 * D, Din), generator=g) * Din ** -0.5).half().float()
bqkv = torch.randn((3 * D,), generator=g) * 0.2
wo = (torch.randn((D, D), generator=g) * D ** -0.5).half().float()
bo = torch.randn((D,), generator=g) * 0.2
packed = tuple(torch.from_numpy(a).to(dev) for a in sam.pack_hiera_attn_pool(wsc.numpy(), bsc.numpy(), wqkv.numpy(), bqkv.numpy(), wo.numpy(), bo.numpy(), heads))
ref = K.hiera_attn_pool(h, packed, n, G, G, heads, D).clone()
torch.cuda.synchronize()
a = torch.randn((65536, 448), device=dev).half()
w = torch.randn((1792, 448), device=dev).half()
s2 = torch.cuda.Stream()
bad = 0
for it in range(30):
    with torch.cuda.stream(s2):
        for _ in range(4):
            K.gemm(a, w, act=K.ACT_GELU)
    out = K.hiera_attn_pool(h, packed, n, G, G, heads, D)
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        d = (out != ref)
        rows = d.any(1).nonzero().flatten()
        bad += 1
        print(f"iteration {it}: {int(d.sum())} elements differ in {rows.numel()} rows; first rows {rows[:8].tolist()}, columns {d[rows[0]].nonzero().flatten()[:12].tolist()}, "
              f"max |diff| {(out - ref).abs().max().item():.3e}")
print("differing iterations:", bad, "of 30")
