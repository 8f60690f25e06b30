"""Fused LN + MLP at D = 448 (Hiera-B+ stage 3; csrc/mlp.hip) against the unfused LayerNorm -> GEMM(GELU) -> GEMM(+res) launches:
bits (the fused kernel keeps the rounding points of the unfused chain: LN output and GELU output in f16, f32 accumulation) and
time at the bench's shape (30 frames x 4096 tokens), with and without the next block's LayerNorm output (h_next)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd"), os.path.join(ROOT, "tools")]
os.environ.setdefault("LMX_MLP448", "1")  # the D = 448 instantiation is a development configuration (csrc/mlp.hip)
from lmx import kernels as K  # noqa: E402
from perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
D = 448
torch.manual_seed(0)
for rows in (1000, 122880):
    x0 = torch.randn((rows, D), device=dev) * 1.5 + 0.3
    g, b = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
    gn, bn = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
    w1 = (torch.randn((4 * D, D), device=dev) * D ** -0.5).half()
    w2 = (torch.randn((D, 4 * D), device=dev) * (4 * D) ** -0.5).half()
    b1, b2 = torch.randn(4 * D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
    xu = x0.clone()
    h = K.layernorm(xu, g, b, 1e-6)
    u = K.gemm(h, w1, bias=b1, act=K.ACT_GELU)
    K.gemm(u, w2, bias=b2, res=xu, out=xu)
    hn_ref = K.layernorm(xu, gn, bn, 1e-6)
    xf = x0.clone()
    hn = torch.empty((rows, D), dtype=torch.float16, device=dev)
    x16 = torch.empty((rows, D), dtype=torch.float16, device=dev)
    K.ln_mlp(xf, g, b, w1, b1, w2, b2, 1e-6, x16=x16, next_ln=(gn, bn, hn))
    torch.cuda.synchronize()
    d = (xf - xu).abs().max().item()
    print(f"rows={rows}: fused vs unfused max |diff| {d:.3e} (|x| max {xu.abs().max().item():.2f}), bit-equal {torch.equal(xf, xu)}; "
          f"h_next max diff {(hn.float() - hn_ref.float()).abs().max().item():.3e}; x16 == cast(x): {torch.equal(x16, xf.half())}", flush=True)
rows = 122880
x = torch.randn((rows, D), device=dev)


def unfused():
    h = K.layernorm(x, g, b, 1e-6)
    u = K.gemm(h, w1, bias=b1, act=K.ACT_GELU)
    K.gemm(u, w2, bias=b2, res=x, out=x)
    K.layernorm(x, gn, bn, 1e-6)


t_u = timeit(unfused, iters=10)
x.normal_()
hn = torch.empty((rows, D), dtype=torch.float16, device=dev)
t_f = timeit(lambda: K.ln_mlp(x, g, b, w1, b1, w2, b2, 1e-6, next_ln=(gn, bn, hn)), iters=10)
x.normal_()
t_f0 = timeit(lambda: K.ln_mlp(x, g, b, w1, b1, w2, b2, 1e-6), iters=10)
fl = 16.0 * rows * D * D
print(f"rows={rows} D={D}: unfused LN+fc1+fc2+nextLN {t_u:.3f} ms | fused with h_next {t_f:.3f} ms ({fl / t_f / 1e9:.0f} TFLOP/s) | fused alone {t_f0:.3f} ms ({fl / t_f0 / 1e9:.0f} TFLOP/s)")
