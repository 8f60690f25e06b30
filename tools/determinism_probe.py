"""Development aid: run each kernel twice on the same inputs and compare bit for bit (a race shows up as a mismatch)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def check(name, fn, reps=6):
    ref = fn()
    torch.cuda.synchronize()
    bad = 0
    worst = 0.0
    for _ in range(reps):
        out = fn()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out.float() - ref.float()).abs().max()))
    print(f"{name}: {'DETERMINISTIC' if not bad else f'{bad}/{reps} runs differ, max abs diff {worst:.4g}'}", flush=True)


for (M, N, K_, act, f32) in [(65536, 1792, 448, 2, 0), (65536, 448, 1792, 0, 1), (65536, 1344, 448, 0, 0), (65536, 448, 448, 0, 1),
                             (16384, 896, 3584, 0, 1), (6432, 1024, 4096, 0, 1), (6432, 4096, 1024, 2, 0), (262144, 224, 896, 0, 1),
                             (1048576, 336, 112, 0, 0)]:
    a = torch.randn((M, K_), device=dev).half()
    w = (torch.randn((N, K_), device=dev) * K_ ** -0.5).half()
    b = torch.randn((N,), device=dev)
    if f32:
        res = torch.randn((M, N), device=dev)
        check(f"gemm {M}x{N}x{K_} f32+res", lambda: K.gemm(a, w, bias=b, act=act, res=res, out_dtype=torch.float32))
    else:
        check(f"gemm {M}x{N}x{K_} f16 act{act}", lambda: K.gemm(a, w, bias=b, act=act))
for rows, D in [(1048576, 112), (262144, 224)]:
    x0 = torch.randn((rows, D), device=dev)
    g, bb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    w1 = (torch.randn((4 * D, D), device=dev) * D ** -0.5).half()
    w2 = (torch.randn((D, 4 * D), device=dev) * (4 * D) ** -0.5).half()
    b1, b2 = torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev)
    check(f"ln_mlp D={D}", lambda: K.ln_mlp(x0.clone(), g, bb, w1, b1, w2, b2, 1e-6))
for (B, H, T, hd, window, nimg) in [(16, 8, 4096, 56, None, None), (400, 8, 196, 56, dict(Gh=64, Gw=64, ws=14, q_stride=1), 16),
                                    (16384, 2, 64, 56, dict(Gh=256, Gw=256, ws=8, q_stride=1), 16),
                                    (16384, 4, 16, 56, dict(Gh=128, Gw=128, ws=4, q_stride=1), 16), (64, 16, 201, 64, None, None)]:
    D = H * hd
    rows = B * T if window is None else nimg * window["Gh"] * window["Gw"]
    qkv = torch.randn((rows, 3 * D), device=dev).half()
    pad = torch.randn((3 * D,), device=dev).half()

    def attn():
        o = torch.zeros((rows, D), device=dev, dtype=torch.float16)
        K.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, T, T, hd, hd ** -0.5, window=window, pad_k=pad[D:2 * D],
                    pad_v=pad[2 * D:])
        return o

    check(f"attention B{B} H{H} T{T} hd{hd} {'win' if window else 'flat'}", attn)
x = torch.randn((131072, 448), device=dev)
g, bb = torch.ones(448, device=dev), torch.zeros(448, device=dev)
check("layernorm 131072x448", lambda: K.layernorm(x, g, bb, 1e-6))
