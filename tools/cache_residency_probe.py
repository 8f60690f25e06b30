"""Development aid: does a consumer find its producer's output in the Infinity Cache?  LayerNorm writes h (f16 [M, 448]), the qkv
GEMM (N = 1344) reads it; buffer sets rotate so that h can only be cache-resident because the LayerNorm just wrote it.
Per M: the GEMM (a) in a loop on one buffer set (hot), (b) rotating over sets beyond the cache (cold), (c) behind its LayerNorm."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
N, Kd = 1344, 448
w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).half()
b = torch.randn(N, device=dev, generator=g)
gam, bet = torch.ones(Kd, device=dev), torch.zeros(Kd, device=dev)
for M in (8192, 16384, 32768, 65536, 122880):
    NS = max(4, min(64, -(-1_200_000_000 // (M * Kd * 2))))
    xs = [torch.randn(M, Kd, device=dev, generator=g) for _ in range(NS)]
    hs = [x.half() for x in xs]
    outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(min(NS, 8))]

    def timed(fn, n=3 * NS):
        fn(0, None)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i in range(n):
            fn(i, ev[i])
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b_) * 1000 for a, b_ in ev)
        return ts[len(ts) // 2]

    def gemm(i, ev, rot):
        k = i % NS if rot else 0
        if ev:
            ev[0].record()
        K.gemm(hs[k], w, bias=b, out=outs[k % len(outs)])
        if ev:
            ev[1].record()

    def ln_gemm(i, ev):
        k = i % NS
        K.layernorm(xs[k], gam, bet, 1e-6, out=hs[k])
        if ev:
            ev[0].record()
        K.gemm(hs[k], w, bias=b, out=outs[k % len(outs)])
        if ev:
            ev[1].record()

    hot, cold, after = timed(lambda i, ev: gemm(i, ev, False)), timed(lambda i, ev: gemm(i, ev, True)), timed(ln_gemm)
    print(f"M={M:7d} (h = {M * Kd * 2 / 1e6:6.1f} MB, {NS} sets): hot {hot:7.1f} us   cold {cold:7.1f} us   behind its LayerNorm {after:7.1f} us", flush=True)
    del xs, hs, outs
