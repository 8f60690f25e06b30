"""lmx_k_hiera_attn8 (csrc/hiera.hip) against the four launches it replaces at Hiera-B+ stage 1's shape in the benched step:
30 frames x 256 x 256 tokens, D = 112, 2 heads, 8 x 8 windows."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import sam  # noqa: E402
from perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
D, heads, hd = 112, 2, 56
for n, G in ((30, 256), (10, 256)):
    rows = n * G * G
    g = torch.Generator().manual_seed(5)
    x = torch.randn((rows, D), generator=g).to(dev)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    wqkv = (torch.randn((3 * D, D), generator=g) * D ** -0.5).half().float()
    bqkv = torch.randn((3 * D,), generator=g) * 0.2
    wo = (torch.randn((D, D), generator=g) * D ** -0.5).half().float()
    bo = torch.randn((D,), generator=g) * 0.2
    packed = tuple(torch.from_numpy(a).to(dev) for a in sam.pack_hiera_attn(wqkv.numpy(), bqkv.numpy(), wo.numpy(), bo.numpy(), heads, ln_inside=True))
    packed_h = tuple(torch.from_numpy(a).to(dev) for a in sam.pack_hiera_attn(wqkv.numpy(), bqkv.numpy(), wo.numpy(), bo.numpy(), heads))
    w16, wo16, bq, bod = wqkv.half().to(dev), wo.half().to(dev), bqkv.to(dev), bo.to(dev)
    a = torch.empty((rows, D), dtype=torch.float16, device=dev)
    pk, pv = torch.zeros(D, dtype=torch.float16, device=dev), torch.zeros(D, dtype=torch.float16, device=dev)

    def unfused():
        h = K.layernorm(x, gam, bet, 1e-6)
        q3 = K.gemm(h, w16, bias=bq)
        K.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], a, n * (G // 8) ** 2, heads, 64, 64, hd, hd ** -0.5,
                    window=dict(Gh=G, Gw=G, ws=8, q_stride=1), pad_k=pk, pad_v=pv)
        K.gemm(a, wo16, bias=bod, res=x, out=x)

    t_u = timeit(unfused, iters=5)
    x.normal_()
    t_f = timeit(lambda: K.hiera_attn8(x, packed, n, G, G, heads, ln=(gam, bet, 1e-6)), iters=5)
    x.normal_()
    hh = K.layernorm(x, gam, bet, 1e-6)
    t_l = timeit(lambda: K.layernorm(x, gam, bet, 1e-6), iters=5)
    t_h = timeit(lambda: K.hiera_attn8(x, packed_h, n, G, G, heads, h=hh), iters=5)
    print(f"{n} frames: LayerNorm launch {t_l * 1e3:.0f} us; fused on precomputed LayerNorm rows {t_h * 1e3:.0f} us", flush=True)
    print(f"{n} frames: unfused (LayerNorm + qkv GEMM + window attention + proj GEMM) {t_u * 1e3:.0f} us, fused incl. LayerNorm {t_f * 1e3:.0f} us "
          f"({rows * 896 / t_f / 1e9:.2f} TB/s algorithmic, {rows / 64 * 624 * 16384 / t_f / 1e9:.0f} TFLOP/s issued)", flush=True)

# stage 2: D = 224, 4 heads, 4 x 4 windows, 30 frames x 128 x 128 tokens (lmx_k_hiera_attn4: weights streamed)
D, heads, hd = 224, 4, 56
for n, G in ((30, 128), (10, 128)):
    rows = n * G * G
    g = torch.Generator().manual_seed(6)
    h = torch.randn((rows, D), generator=g).half().to(dev)
    x = torch.randn((rows, D), generator=g).to(dev)
    wqkv = (torch.randn((3 * D, D), generator=g) * D ** -0.5).half().float()
    bqkv = torch.randn((3 * D,), generator=g) * 0.2
    wo = (torch.randn((D, D), generator=g) * D ** -0.5).half().float()
    bo = torch.randn((D,), generator=g) * 0.2
    packed4 = tuple(torch.from_numpy(a).to(dev) for a in sam.pack_hiera_attn4(wqkv.numpy(), bqkv.numpy(), wo.numpy(), bo.numpy(), heads))
    w16, wo16, bq, bod = wqkv.half().to(dev), wo.half().to(dev), bqkv.to(dev), bo.to(dev)
    a = torch.empty((rows, D), dtype=torch.float16, device=dev)
    pk, pv = torch.zeros(D, dtype=torch.float16, device=dev), torch.zeros(D, dtype=torch.float16, device=dev)

    def unfused4():
        q3 = K.gemm(h, w16, bias=bq)
        K.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], a, n * (G // 4) ** 2, heads, 16, 16, hd, hd ** -0.5,
                    window=dict(Gh=G, Gw=G, ws=4, q_stride=1), pad_k=pk, pad_v=pv)
        K.gemm(a, wo16, bias=bod, res=x, out=x)

    t_u = timeit(unfused4, iters=5)
    x.normal_()
    t_f = timeit(lambda: K.hiera_attn4(h, x, packed4, n, G, G, heads), iters=5)
    print(f"stage 2, {n} frames: unfused (qkv GEMM + window attention + proj GEMM) {t_u * 1e3:.0f} us, fused {t_f * 1e3:.0f} us "
          f"({rows * 2240 / t_f / 1e9:.2f} TB/s algorithmic)", flush=True)

# the block that opens stage 2: 112 -> 224 channels, 8 x 8 windows, pooled queries and shortcut (lmx_k_hiera_attn_pool)
Din, D, heads, hd = 112, 224, 4, 56
for n, G in ((30, 256), (10, 256)):
    rows = n * G * G
    g = torch.Generator().manual_seed(7)
    h = torch.randn((rows, Din), generator=g).half().to(dev)
    wsc = (torch.randn((D, Din), generator=g) * Din ** -0.5).half().float()
    bsc = torch.randn((D,), generator=g) * 0.2
    wqkv = (torch.randn((3 * D, Din), generator=g) * Din ** -0.5).half().float()
    bqkv = torch.randn((3 * D,), generator=g) * 0.2
    wo = (torch.randn((D, D), generator=g) * D ** -0.5).half().float()
    bo = torch.randn((D,), generator=g) * 0.2
    packedp = tuple(torch.from_numpy(a).to(dev) for a in sam.pack_hiera_attn_pool(wsc.numpy(), bsc.numpy(), wqkv.numpy(), bqkv.numpy(), wo.numpy(),
                                                                                   bo.numpy(), heads))
    wsc16, w16, wo16 = wsc.half().to(dev), wqkv.half().to(dev), wo.half().to(dev)
    bscd, bq, bod = bsc.to(dev), bqkv.to(dev), bo.to(dev)
    a = torch.empty((rows // 4, D), dtype=torch.float16, device=dev)
    xo = torch.empty((rows // 4, D), dtype=torch.float32, device=dev)
    pk, pv = torch.zeros(D, dtype=torch.float16, device=dev), torch.zeros(D, dtype=torch.float16, device=dev)

    def unfusedp():  # as HieraEncoder._attention_half runs it (pooled GEMM forms)
        sc = K.gemm(h, wsc16, bias=bscd, out_dtype=torch.float32, pool_hw=(G, G))
        q = K.gemm(h, w16[:D], bias=bq[:D], pool_hw=(G, G))
        kv = K.gemm(h, w16[D:], bias=bq[D:])
        K.attention(q, kv[:, :D], kv[:, D:], a, n * (G // 8) ** 2, heads, 16, 64, hd, hd ** -0.5,
                    window=dict(Gh=G, Gw=G, ws=8, q_stride=2), pad_k=pk, pad_v=pv)
        K.gemm(a, wo16, bias=bod, res=sc, out=xo)

    t_u = timeit(unfusedp, iters=5)
    t_f = timeit(lambda: K.hiera_attn_pool(h, packedp, n, G, G, heads, D), iters=5)
    print(f"stage-opening block, {n} frames: unfused (shortcut, q, kv GEMMs + window attention + proj GEMM) {t_u * 1e3:.0f} us, fused {t_f * 1e3:.0f} us "
          f"({(rows * 224 + rows // 4 * 896) / t_f / 1e9:.2f} TB/s algorithmic)", flush=True)

# the block that opens stage 3: 224 -> 448 channels, 4 x 4 windows, pooled queries and shortcut
Din, D, heads, hd = 224, 448, 8, 56
for n, G in ((30, 128), (10, 128)):
    rows = n * G * G
    g = torch.Generator().manual_seed(8)
    h = torch.randn((rows, Din), generator=g).half().to(dev)
    wsc = (torch.randn((D, Din), generator=g) * Din ** -0.5).half().float()
    bsc = torch.randn((D,), generator=g) * 0.2
    wqkv = (torch.randn((3 * D, Din), generator=g) * Din ** -0.5).half().float()
    bqkv = torch.randn((3 * D,), generator=g) * 0.2
    wo = (torch.randn((D, D), generator=g) * D ** -0.5).half().float()
    bo = torch.randn((D,), generator=g) * 0.2
    packedq = tuple(torch.from_numpy(a).to(dev) for a in sam.pack_hiera_attn_pool(wsc.numpy(), bsc.numpy(), wqkv.numpy(), bqkv.numpy(), wo.numpy(),
                                                                                   bo.numpy(), heads))
    wsc16, w16, wo16 = wsc.half().to(dev), wqkv.half().to(dev), wo.half().to(dev)
    bscd, bq, bod = bsc.to(dev), bqkv.to(dev), bo.to(dev)
    a = torch.empty((rows // 4, D), dtype=torch.float16, device=dev)
    xo = torch.empty((rows // 4, D), dtype=torch.float32, device=dev)
    pk, pv = torch.zeros(D, dtype=torch.float16, device=dev), torch.zeros(D, dtype=torch.float16, device=dev)

    def unfusedq():
        sc = K.gemm(h, wsc16, bias=bscd, out_dtype=torch.float32, pool_hw=(G, G))
        q = K.gemm(h, w16[:D], bias=bq[:D], pool_hw=(G, G))
        kv = K.gemm(h, w16[D:], bias=bq[D:])
        K.attention(q, kv[:, :D], kv[:, D:], a, n * (G // 4) ** 2, heads, 4, 16, hd, hd ** -0.5,
                    window=dict(Gh=G, Gw=G, ws=4, q_stride=2), pad_k=pk, pad_v=pv)
        K.gemm(a, wo16, bias=bod, res=sc, out=xo)

    t_u = timeit(unfusedq, iters=5)
    t_f = timeit(lambda: K.hiera_attn_pool(h, packedq, n, G, G, heads, D), iters=5)
    print(f"stage-3 opener, {n} frames: unfused (shortcut, q, kv GEMMs + window attention + proj GEMM) {t_u * 1e3:.0f} us, fused {t_f * 1e3:.0f} us "
          f"({(rows * 448 + rows // 4 * 1792) / t_f / 1e9:.2f} TB/s algorithmic)", flush=True)
