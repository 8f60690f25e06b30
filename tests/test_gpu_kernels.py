"""Per-kernel parity of liblmx (through the C-ABI) against plain fp32 CPU references on the same seeded inputs.
Tolerances: integer/byte/index kernels bit-exact; f16-output kernels within f16 rounding of the fp32 result."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _close(got, ref, atol, rtol, what):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} at ref {float(ref.flatten()[err.argmax()]):.4g}"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (201, 1024, 1024), (77, 80, 72), (1000, 336, 112), (513, 64, 2048),
                                   (4096, 3072, 1024), (777, 200, 152), (1111, 448, 224), (512, 96, 8), (2049, 1792, 448)])
@pytest.mark.parametrize("out_dtype", [torch.float16, torch.float32])
def test_gemm_plain(cuda, M, N, K, out_dtype):
    from lmx import kernels as Kk

    a = _rand((M, K), 1).half()
    w = _rand((N, K), 2, K ** -0.5).half()
    b = _rand((N,), 3)
    ref = a.float() @ w.float().t() + b
    got = Kk.gemm(a.to(cuda), w.to(cuda), bias=b.to(cuda), out_dtype=out_dtype)
    tol = dict(atol=2e-3, rtol=2e-3) if out_dtype == torch.float16 else dict(atol=2e-4, rtol=1e-4)
    _close(got, ref, what=f"gemm {M}x{N}x{K}", **tol)


@pytest.mark.parametrize("M", [300, 1500])  # 300 -> 128x128 register-staged kernel, 1500 -> 256x128 LDS-DMA kernel
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogues(cuda, act, M):
    from lmx import kernels as Kk

    N, K = 256, 192
    a = _rand((M, K), 4).half()
    w = _rand((N, K), 5, K ** -0.5).half()
    b, s = _rand((N,), 6), _rand((N,), 7)
    res = _rand((M, N), 8)
    y = a.float() @ w.float().t() + b
    y = [y, F.silu(y), F.gelu(y), F.relu(y)][act]
    ref = y * s + res
    # f32 residual stream updated in place through a strided view (ld > N)
    buf = torch.zeros((M, N + 64), dtype=torch.float32)
    buf[:, :N] = res
    buf = buf.to(cuda)
    view = buf[:, :N]
    Kk.gemm(a.to(cuda), w.to(cuda), bias=b.to(cuda), act=act, scale=s.to(cuda), res=view, out=view)
    _close(view, ref, 3e-4, 1e-4, f"gemm epilogue act={act}")
    assert float(buf[:, N:].abs().max()) == 0.0
    # f16 in/out with f16 residual
    resh = res.half()
    ref16 = y * s + resh.float()
    got = Kk.gemm(a.to(cuda), w.to(cuda), bias=b.to(cuda), act=act, scale=s.to(cuda), res=resh.to(cuda),
                  out_dtype=torch.float16)
    _close(got, ref16, 4e-3, 2e-3, f"gemm f16 epilogue act={act}")


@pytest.mark.parametrize("n,H,W,K,N", [(3, 20, 20, 112, 224), (2, 64, 64, 224, 448), (5, 16, 16, 448, 896), (1, 30, 26, 72, 104)])
def test_gemm_pooled_rows_equal_gemm_then_maxpool(cuda, n, H, W, K, N):
    """a_mode 2 (Hiera's do_pool(proj(x)) in one launch): bit for bit the f32 GEMM followed by maxpool2, ragged last tile,
    several images, N not a multiple of the tile width."""
    from lmx import kernels as Kk

    M = n * H * W
    a = _rand((M, K), 21).half().to(cuda)
    w = _rand((N, K), 22, K ** -0.5).half().to(cuda)
    b = _rand((N,), 23).to(cuda)
    full = Kk.gemm(a, w, bias=b, out_dtype=torch.float32)
    ref = torch.empty((n, H // 2, W // 2, N), dtype=torch.float32, device=cuda)
    Kk.maxpool2(full.view(n, H, W, N), ref)
    got = Kk.gemm(a, w, bias=b, out_dtype=torch.float32, pool_hw=(H, W))
    assert got.shape == (M // 4, N)
    assert torch.equal(got, ref.view(-1, N))
    # f16 output (Hiera's pooled queries): the f16 GEMM followed by the f16 max-pool
    full16 = Kk.gemm(a, w, bias=b)
    ref16 = torch.empty((n, H // 2, W // 2, N), dtype=torch.float16, device=cuda)
    Kk.maxpool2(full16.view(n, H, W, N), ref16)
    assert torch.equal(Kk.gemm(a, w, bias=b, pool_hw=(H, W)), ref16.view(-1, N))
    # and against plain fp32 arithmetic
    want = F.max_pool2d((a.float() @ w.float().t() + b).view(n, H, W, N).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(-1, N)
    _close(got, want, 2e-4, 1e-4, "pooled gemm")
    with pytest.raises(Exception):
        Kk.gemm(a[:64], w, bias=b, out_dtype=torch.float32, pool_hw=(8, 8))  # below the LDS-DMA kernel's sizes: refused loudly


@pytest.mark.parametrize("n,H,W,Cin,Cout,stride", [(2, 20, 20, 64, 128, 1), (1, 33, 47, 16, 24, 2), (2, 40, 24, 128, 64, 2),
                                                    (1, 80, 80, 256, 256, 1)])
def test_conv3x3(cuda, n, H, W, Cin, Cout, stride):
    from lmx import kernels as Kk

    x = _rand((n, H, W, Cin), 10).half()
    w = _rand((Cout, Cin, 3, 3), 11, (9 * Cin) ** -0.5).half()
    b = _rand((Cout,), 12, 0.1)
    ref = F.silu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=stride, padding=1)).permute(0, 2, 3, 1)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    got = Kk.conv3x3(x.to(cuda), wp.to(cuda), bias=b.to(cuda), act=Kk.ACT_SILU, stride=stride)
    _close(got, ref, 3e-3, 3e-3, f"conv3x3 {n}x{H}x{W}x{Cin}->{Cout} s{stride}")


def test_conv_channel_slices_and_residual(cuda):
    """C2f plumbing: read a channel slice, write another slice of the same buffer, add the shortcut."""
    from lmx import kernels as Kk

    n, H, W, C = 2, 16, 12, 32
    buf = _rand((n, H, W, 4 * C), 13).half()
    w = _rand((C, C, 3, 3), 14, (9 * C) ** -0.5).half()
    b = _rand((C,), 15, 0.1)
    xin = buf[..., C:2 * C]
    ref = buf.clone().float()
    y = F.silu(F.conv2d(xin.float().permute(0, 3, 1, 2), w.float(), b, padding=1)).permute(0, 2, 3, 1)
    ref[..., 2 * C:3 * C] = y + xin.float()
    d = buf.to(cuda)
    wp = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().to(cuda)
    Kk.conv3x3(d[..., C:2 * C], wp, bias=b.to(cuda), res=d[..., C:2 * C], out=d[..., 2 * C:3 * C])
    _close(d, ref, 4e-3, 3e-3, "conv slice/residual")
    # 1x1 over the whole concat buffer into a fresh tensor
    w1 = _rand((48, 4 * C), 16, (4 * C) ** -0.5).half()
    ref1 = F.silu(ref.half().float() @ w1.float().t())
    got1 = Kk.conv1x1(d, w1.to(cuda), act=Kk.ACT_SILU)
    _close(got1, ref1, 5e-3, 5e-3, "conv1x1")


def test_conv_lds_dma_path_slices_residual_stride2(cuda):
    """Shapes that take the 256x128 LDS-DMA kernel (M >= 512, Cout >= 96, Cin % 32 == 0): slice in/out, shortcut, stride 2,
    a tile spanning two images, M not a multiple of the tile."""
    from lmx import kernels as Kk

    n, H, W, C = 3, 24, 20, 128
    buf = _rand((n, H, W, 3 * C), 17).half()
    w = _rand((C, C, 3, 3), 18, (9 * C) ** -0.5).half()
    b = _rand((C,), 19, 0.1)
    xin = buf[..., C:2 * C]
    ref = buf.clone().float()
    y = F.silu(F.conv2d(xin.float().permute(0, 3, 1, 2), w.float(), b, padding=1)).permute(0, 2, 3, 1)
    ref[..., 2 * C:] = y + xin.float()
    d = buf.to(cuda)
    wp = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().to(cuda)
    Kk.conv3x3(d[..., C:2 * C], wp, bias=b.to(cuda), res=d[..., C:2 * C], out=d[..., 2 * C:])
    _close(d, ref, 4e-3, 3e-3, "lds-dma conv slice/residual")
    x2 = _rand((2, 47, 33, 64), 20).half()
    w2 = _rand((96, 64, 3, 3), 21, (9 * 64) ** -0.5).half()
    ref2 = F.silu(F.conv2d(x2.float().permute(0, 3, 1, 2), w2.float(), None, stride=2, padding=1)).permute(0, 2, 3, 1)
    got2 = Kk.conv3x3(x2.to(cuda), w2.permute(0, 2, 3, 1).reshape(96, 576).contiguous().to(cuda), act=Kk.ACT_SILU, stride=2)
    _close(got2, ref2, 4e-3, 3e-3, "lds-dma conv stride 2")


# ------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,D", [(5, 112), (201, 1024), (64, 1280), (3, 4096)])
def test_layernorm(cuda, rows, D):
    from lmx import kernels as Kk

    x = _rand((rows, D), 20, 3.0) + 0.5
    g, b = _rand((D,), 21) * 0.1 + 1, _rand((D,), 22) * 0.1
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    got = Kk.layernorm(x.to(cuda), g.to(cuda), b.to(cuda), 1e-5, out_dtype=torch.float32)
    _close(got, ref, 2e-5, 2e-5, "layernorm f32")
    got16 = Kk.layernorm(x.to(cuda), g.to(cuda), b.to(cuda), 1e-5)
    _close(got16, ref, 2e-3, 1e-3, "layernorm f16 out")
    got_h = Kk.layernorm(x.half().to(cuda), g.to(cuda), b.to(cuda), 1e-5, out_dtype=torch.float32)
    _close(got_h, F.layer_norm(x.half().float(), (D,), g, b, 1e-5), 2e-5, 2e-5, "layernorm f16 in")


# ------------------------------------------------------------------------ fused LayerNorm + MLP (narrow widths)
@pytest.mark.parametrize("rows,D", [(1, 112), (130, 112), (4099, 112), (63, 224), (2050, 224)])
def test_ln_mlp_fused(cuda, rows, D):
    """x += fc2(gelu(fc1(LN(x)))) in one kernel: against fp32 torch on the f16-rounded weights, and against the unfused
    LN -> GEMM -> GEMM launches of the same library (same rounding points: only the accumulation order differs)."""
    from lmx import kernels as Kk

    x = _rand((rows, D), 70, 2.0) + 0.3
    g, b = _rand((D,), 71) * 0.1 + 1, _rand((D,), 72) * 0.1
    w1 = (_rand((4 * D, D), 73) * D ** -0.5).half()
    w2 = (_rand((D, 4 * D), 74) * (4 * D) ** -0.5).half()
    b1, b2 = _rand((4 * D,), 75) * 0.1, _rand((D,), 76) * 0.1
    ref = x + F.linear(F.gelu(F.linear(F.layer_norm(x, (D,), g, b, 1e-6), w1.float(), b1)), w2.float(), b2)
    dev = [t.to(cuda) for t in (g, b, w1, b1, w2, b2)]
    got = Kk.ln_mlp(x.clone().to(cuda), *dev, 1e-6)
    _close(got, ref, 6e-3, 3e-3, "fused ln_mlp vs fp32")
    xu = x.clone().to(cuda)
    h = Kk.layernorm(xu, dev[0], dev[1], 1e-6)
    u = Kk.gemm(h, dev[2], bias=dev[3], act=Kk.ACT_GELU)
    Kk.gemm(u, dev[4], bias=dev[5], res=xu, out=xu)
    _close(got, xu, 2e-3, 1e-3, "fused ln_mlp vs unfused launches")
    # the optional f16 copy of the result is what a cast of the f32 result gives, and asking for it changes nothing else
    x16 = torch.full((rows, D), 7.0, dtype=torch.float16, device=cuda)
    got2 = Kk.ln_mlp(x.clone().to(cuda), *dev, 1e-6, x16=x16)
    assert torch.equal(got2, got) and torch.equal(x16, Kk.cast_f16(got))
    # the next block's LayerNorm from the same launch: the rows' statistics are reduced in another order than the LayerNorm
    # kernel's, so equality is to f16 rounding, not to the bit
    gn, bn = (_rand((D,), 77) * 0.1 + 1).to(cuda), (_rand((D,), 78) * 0.1).to(cuda)
    hn = torch.full((rows, D), 7.0, dtype=torch.float16, device=cuda)
    got3 = Kk.ln_mlp(x.clone().to(cuda), *dev, 1e-6, next_ln=(gn, bn, hn))
    assert torch.equal(got3, got)
    _close(hn, Kk.layernorm(got, gn, bn, 1e-6), 2e-3, 2e-3, "next-block LayerNorm from the fused MLP")
    _close(hn, F.layer_norm(got.float().cpu(), (D,), gn.cpu(), bn.cpu(), 1e-6), 2e-3, 2e-3, "next-block LayerNorm vs torch")


def test_ln_mlp_rejects_other_widths(cuda):
    from lmx import kernels as Kk
    from lmx._lib import LmxError

    D = 448
    z = torch.zeros
    with pytest.raises(LmxError):
        Kk.ln_mlp(z((8, D), device=cuda), z(D, device=cuda), z(D, device=cuda), z((4 * D, D), device=cuda, dtype=torch.float16),
                  z(4 * D, device=cuda), z((D, 4 * D), device=cuda, dtype=torch.float16), z(D, device=cuda), 1e-6)


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale):
    w = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    return w @ v


# T >= 256 takes the LDS-DMA ring (attn.hip DMA path): ragged last tile, head-dim padding chunks, one / many key tiles
@pytest.mark.parametrize("B,H,T,hd", [(2, 16, 201, 64), (3, 2, 64, 56), (1, 4, 300, 32), (2, 3, 16, 64), (1, 2, 1000, 56),
                                      (2, 3, 257, 64), (3, 2, 256, 56), (1, 2, 4096, 56), (2, 1, 1153, 48),
                                      # (the pipelined form attn_gp_kernel: odd / even tile counts, full and ragged last tile, dot2 sums)
                                      (1, 4, 320, 64), (1, 1, 4032, 56), (2, 2, 258, 40), (1, 2, 384, 64),
                                      # 128 < T <= 208 takes the whole-sequence kernel (attn_sp_kernel): both edges, a ragged
                                      # last 16-key block, an odd number of 16-query blocks, head-dim padding, ones-column / dot2 sums
                                      (2, 8, 196, 56), (1, 3, 208, 64), (2, 2, 129, 32), (3, 5, 150, 48), (1, 2, 193, 64), (2, 1, 209, 56),
                                      # >= 64 (batch, head) items of such a shape take the PERSISTENT form (attn_spp_kernel: LDS-DMA double
                                      # buffer, one workgroup walks several items — more items than CUs in the last case, an odd count per XCD)
                                      (9, 8, 196, 56), (8, 16, 201, 64), (22, 3, 129, 32), (13, 5, 150, 48), (70, 4, 208, 64), (37, 9, 193, 40),
                                      # head dims 72..96 take the wide class (128-half LDS rows, 3 k-steps, 6 output blocks)
                                      (2, 3, 201, 80), (1, 2, 64, 96), (2, 2, 15, 72), (1, 1, 700, 88)])
def test_attention_flat(cuda, B, H, T, hd):
    from lmx import kernels as Kk

    D = H * hd
    qkv = _rand((B * T, 3 * D), 30, 1.5).half()
    q, k, v = (qkv[:, i * D:(i + 1) * D].float().view(B, T, H, hd).transpose(1, 2) for i in range(3))
    ref = _attn_ref(q, k, v, hd ** -0.5).transpose(1, 2).reshape(B * T, D)
    d = qkv.to(cuda)
    out = torch.zeros((B * T, D), dtype=torch.float16, device=cuda)
    Kk.attention(d[:, :D], d[:, D:2 * D], d[:, 2 * D:], out, B, H, T, T, hd, hd ** -0.5)
    _close(out, ref, 3e-3, 3e-3, f"attention B{B} H{H} T{T} hd{hd}")


def _window_ref(x_q, x_k, x_v, Gh, Gw, ws, heads, hd, pad_k, pad_v, q_stride=1):
    """window_partition with zero padding replaced by the qkv-bias rows, attention per window, unpartition."""
    n = x_k.shape[0] // (Gh * Gw)
    D = heads * hd

    def part(x, padrow, gh, gw, w):
        x = x.view(n, gh, gw, D)
        php, pwp = (-gh) % w, (-gw) % w
        if php or pwp:
            full = padrow.view(1, 1, 1, D).expand(n, gh + php, gw + pwp, D).clone()
            full[:, :gh, :gw] = x
            x = full
        Hp, Wp = x.shape[1], x.shape[2]
        x = x.view(n, Hp // w, w, Wp // w, w, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, w * w, D)
        return x, (Hp, Wp)

    kx, _ = part(x_k, pad_k, Gh, Gw, ws)
    vx, _ = part(x_v, pad_v, Gh, Gw, ws)
    wq = ws // q_stride
    qx, (Hq, Wq) = part(x_q, torch.zeros(D), Gh // q_stride, Gw // q_stride, wq)
    nb = kx.shape[0]
    qh = qx.view(nb, wq * wq, heads, hd).transpose(1, 2)
    kh = kx.view(nb, ws * ws, heads, hd).transpose(1, 2)
    vh = vx.view(nb, ws * ws, heads, hd).transpose(1, 2)
    o = _attn_ref(qh, kh, vh, hd ** -0.5).transpose(1, 2).reshape(nb, wq * wq, D)
    o = o.view(n, Hq // wq, Wq // wq, wq, wq, D).permute(0, 1, 3, 2, 4, 5).reshape(n, Hq, Wq, D)
    return o[:, :Gh // q_stride, :Gw // q_stride].reshape(-1, D)


@pytest.mark.parametrize("n,Gh,Gw,ws,heads,hd,qs", [(2, 16, 16, 8, 2, 56, 1), (1, 20, 20, 14, 4, 56, 1), (2, 12, 12, 4, 3, 64, 1),
                                                    (1, 64, 64, 14, 2, 64, 1), (1, 16, 16, 8, 4, 56, 2), (1, 20, 20, 14, 2, 56, 2),
                                                    # 4 x 4 windows take the one-wave-per-window kernel: padded grids, Q-pool
                                                    # (4 queries), an item count that is not a multiple of 4
                                                    (1, 12, 12, 4, 2, 56, 2), (2, 10, 10, 4, 5, 56, 1), (1, 8, 8, 4, 1, 32, 1),
                                                    (1, 20, 20, 14, 2, 80, 1), (1, 12, 12, 4, 2, 80, 1),
                                                    # 14 x 14 windows with >= 64 (window, head) items: the persistent kernel, with
                                                    # padded windows on both edges (64 = 4 * 14 + 8; 30 x 44), head-dim padding, hd 64
                                                    (3, 64, 64, 14, 4, 56, 1), (2, 30, 44, 14, 8, 32, 1), (1, 64, 64, 14, 8, 64, 1), (12, 28, 28, 14, 2, 56, 1)])
def test_attention_window(cuda, n, Gh, Gw, ws, heads, hd, qs):
    from lmx import kernels as Kk

    D = heads * hd
    rows = n * Gh * Gw
    qkv = _rand((rows, 3 * D), 31, 1.5).half()
    bias = _rand((3 * D,), 32).half()
    kf, vf = qkv[:, D:2 * D].float(), qkv[:, 2 * D:].float()
    if qs == 1:
        qd = qkv[:, :D].contiguous()
    else:  # Hiera Q-pool: queries live on the 2x-subsampled grid
        qd = _rand((n * (Gh // 2) * (Gw // 2), D), 33, 1.5).half()
    ref = _window_ref(qd.float(), kf, vf, Gh, Gw, ws, heads, hd, bias[D:2 * D].float(), bias[2 * D:].float(), qs)
    d = qkv.to(cuda)
    bd = bias.to(cuda)
    nW = -(-Gh // ws) * -(-Gw // ws)
    out = torch.zeros((qd.shape[0], D), dtype=torch.float16, device=cuda)
    wq = ws // qs
    Kk.attention(qd.to(cuda), d[:, D:2 * D], d[:, 2 * D:], out, n * nW, heads, wq * wq, ws * ws, hd, hd ** -0.5,
                 window=dict(Gh=Gh, Gw=Gw, ws=ws, q_stride=qs), pad_k=bd[D:2 * D], pad_v=bd[2 * D:])
    _close(out, ref, 3e-3, 3e-3, f"window attention {Gh}x{Gw} ws{ws} qs{qs}")


def test_rope(cuda):
    from lmx import dino
    from lmx import kernels as Kk
    from oracle import vit

    cfg = dino.DinoConfig(hidden=128, heads=2)
    B, T, H, hd, npre = 2, 5 + 196, 2, 64, 5
    x = _rand((B * T, 3 * H * hd), 40).half()
    cos, sin = vit.rope_tables(hd, 100.0, 14, 14)
    q = x[:, :H * hd].float().view(B, T, H, hd)
    pat = q[:, npre:]
    rot = torch.cat((-pat[..., hd // 2:], pat[..., :hd // 2]), -1)
    ref = q.clone()
    ref[:, npre:] = pat * cos[None, :, None, :] + rot * sin[None, :, None, :]
    d = x.to(cuda)
    Kk.rope(d[:, :H * hd], B, T, H, hd, npre, cos.to(cuda), sin.to(cuda))
    _close(d[:, :H * hd], ref.reshape(B * T, H * hd), 2e-3, 2e-3, "rope")
    assert torch.equal(d[:, H * hd:].cpu(), x[:, H * hd:])


# --------------------------------------------------------------------------------------- preprocessing
@pytest.mark.parametrize("h,w", [(1080, 1920), (720, 1280), (300, 256)])
def test_dino_preprocess_bit_exact(cuda, h, w):
    """u8 resize must equal Pillow bit for bit; the patch matrix must equal f16(pixel_values) exactly."""
    from lmx import dino, weights
    from oracle import preprocess as OP

    cfg = dino.DinoConfig(hidden=64, layers=1, heads=1, mlp=64)
    emb = dino.DinoEmbedder(cfg, weights.synth_state_dict(dino.param_spec(cfg), 1), cuda)
    frames = np.random.default_rng(50).integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    # smooth one frame so the resampler is not fed pure noise only
    frames[1] = (np.add.outer(np.arange(h), np.arange(w))[:, :, None] // 7 % 256).astype(np.uint8)
    patches = emb.preprocess(torch.from_numpy(frames).to(cuda)).cpu()
    for i in range(2):
        pv = torch.from_numpy(OP.dino_pixel_values(frames[i]))  # [3,224,224]
        ref = pv.view(3, 14, 16, 14, 16).permute(1, 3, 2, 4, 0).reshape(196, 768).half()
        assert torch.equal(patches[i * 196:(i + 1) * 196], ref), f"frame {i} of {h}x{w}"


def test_token_mean_and_assemble(cuda):
    from lmx import kernels as Kk

    B, np_, npre, D = 3, 10, 2, 64
    patch = _rand((B * np_, D), 60).half()
    prefix, pos = _rand((npre, D), 61), _rand((np_ + npre, D), 62)
    ref = torch.cat([prefix.expand(B, -1, -1), patch.float().view(B, np_, D)], 1) + pos
    got = Kk.assemble_tokens(patch.to(cuda), prefix.to(cuda), pos.to(cuda), B, np_, npre, D)
    assert torch.equal(got.cpu().view(B, np_ + npre, D), ref)
    m = Kk.token_mean(got, B, np_ + npre, D)
    _close(m, ref.mean(1), 1e-6, 1e-6, "token_mean")


# ------------------------------------------------------------------------------------------------- NMS
def _run_nms(cuda, pred, conf, iou=0.7, max_det=300):
    from lmx import kernels as Kk

    b, s, c, src, cnt = Kk.nms(torch.from_numpy(pred).to(cuda), conf, iou, max_det)
    torch.cuda.synchronize()
    return b.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), src.cpu().numpy(), cnt.cpu().numpy()


@pytest.mark.parametrize("A,nc,conf,seed", [(8400, 80, 0.25, 0), (8400, 80, 0.001, 1), (5040, 80, 0.5, 2), (300, 3, 0.1, 3),
                                            (16384, 2, 0.05, 4)])
def test_nms_bit_exact(cuda, A, nc, conf, seed):
    from oracle import nms as ON

    rng = np.random.default_rng(seed)
    n = 3
    pred = np.zeros((n, A, 4 + nc), np.float32)
    # clustered boxes so that suppression chains actually occur
    centers = rng.uniform(50, 590, (n, 40, 2))
    which = rng.integers(0, 40, (n, A))
    pred[..., 0:2] = np.take_along_axis(centers, which[..., None].repeat(2, -1), 1) + rng.normal(0, 6, (n, A, 2))
    pred[..., 2:4] = rng.uniform(20, 120, (n, A, 2))
    pred[..., 4:] = rng.uniform(0, 1, (n, A, nc)).astype(np.float32) ** 6
    pred[0, :50, 4:] = pred[0, 50:100, 4:]  # exact score ties between different anchors
    b, s, c, src, cnt = _run_nms(cuda, pred, conf)
    for i in range(n):
        rb, rs, rc, rsrc = ON.non_max_suppression(pred[i], conf)
        k = len(rsrc)
        assert cnt[i] == k, f"image {i}: count {cnt[i]} != {k}"
        assert np.array_equal(src[i, :k], rsrc), f"image {i}: keep set differs"
        assert np.array_equal(c[i, :k], rc)
        assert np.array_equal(s[i, :k], rs)
        assert np.array_equal(b[i, :k], rb)


def test_nms_known_answers(cuda):
    from test_nms_oracle import _pred

    p = _pred([[0, 0, 100, 100], [0, 0, 100, 81], [0, 0, 100, 70], [200, 200, 300, 300]], [0.9, 0.8, 0.85, 0.7], [0, 0, 0, 0])
    _, _, _, src, cnt = _run_nms(cuda, p[None], 0.25)
    assert cnt[0] == 3 and src[0, :3].tolist() == [0, 2, 3]  # IoU 0.81 suppressed, IoU f32(0.7) kept
    n = 400
    p = _pred([[i * 20, 0, i * 20 + 10, 10] for i in range(n)], [0.75] * n, [0] * n)
    _, _, _, src, cnt = _run_nms(cuda, p[None], 0.25)
    assert cnt[0] == 300 and src[0].tolist() == list(range(300))
    p = _pred([[0, 0, 10, 10]], [0.1], [0])
    _, _, _, src, cnt = _run_nms(cuda, p[None], 0.25)
    assert cnt[0] == 0


# ------------------------------------------------------------------------------------------- mask bit-packing
@pytest.mark.parametrize("shape", [(2, 1080, 1920), (1, 7, 13), (3, 5, 8), (1, 1, 1), (2, 33, 250)])
def test_pack_bits_bit_exact(cuda, shape):
    from lmx import kernels as Kk

    m = (np.random.default_rng(90).random(shape) < 0.3).astype(np.uint8)
    m[..., 0] *= 255  # any non-zero byte is a set pixel
    got = Kk.pack_bits(torch.from_numpy(m).to(cuda)).cpu().numpy()
    ref = np.packbits(m != 0, axis=-1)
    assert got.shape == ref.shape and np.array_equal(got, ref)


# ------------------------------------------------------------------------------------------- run-to-run reproducibility
def test_large_launches_are_bit_reproducible(cuda):
    """Every kernel here is meant to be deterministic.  A launch large enough to run several rounds of workgroups per CU,
    repeated on identical inputs, must give identical bits: an earlier version of the fused MLP normalised inside the kernel
    and produced run-to-run different rows (a few hundred per million) only when workgroups were co-resident on a CU —
    invisible to tolerance checks on small inputs (tools/mlp_selfcheck.py, tools/determinism_probe.py)."""
    from lmx import kernels as Kk

    torch.manual_seed(0)
    D, rows = 112, 1 << 20
    x0 = torch.randn((rows, D), device=cuda)
    g, bb = torch.ones(D, device=cuda), torch.zeros(D, device=cuda)
    w1 = (torch.randn((4 * D, D), device=cuda) * D ** -0.5).half()
    w2 = (torch.randn((D, 4 * D), device=cuda) * (4 * D) ** -0.5).half()
    b1, b2 = torch.randn(4 * D, device=cuda) * 0.1, torch.randn(D, device=cuda) * 0.1
    ref = Kk.ln_mlp(x0.clone(), g, bb, w1, b1, w2, b2, 1e-6)
    for _ in range(3):
        assert torch.equal(Kk.ln_mlp(x0.clone(), g, bb, w1, b1, w2, b2, 1e-6), ref), "fused MLP differs between identical launches"
    a = torch.randn((65536, 448), device=cuda).half()
    w = (torch.randn((1792, 448), device=cuda) * 448 ** -0.5).half()
    r0 = Kk.gemm(a, w, act=Kk.ACT_GELU)
    for _ in range(3):
        assert torch.equal(Kk.gemm(a, w, act=Kk.ACT_GELU), r0), "GEMM differs between identical launches"


@pytest.mark.parametrize("n,H,W,cin,cout,stride,S", [(3, 24, 40, 96, 256, 1, 4), (1, 12, 20, 192, 128, 1, 8), (2, 48, 80, 96, 128, 2, 3), (5, 24, 40, 768, 256, 1, 8), (2, 12, 20, 768, 64, 1, 8)])
def test_conv3x3_split_k_partials(cuda, n, H, W, cin, cout, stride, S):
    """lmx_k_gemm split_k (the exact plan's long-K 3 x 3 convolutions): the S partial outputs add up to the unsplit f32 result
    (another summation order: ~1e-6 relative, not bit-equal), bias and scale are applied once, the parts are the same whatever
    the batch (a frame alone gives the bits it gives inside the batch) and lmx_k_split3 adds them in index order."""
    from lmx import kernels as K

    g = torch.Generator(device=cuda).manual_seed(cin + S)
    x = torch.randn((n, H, W, cin), device=cuda, generator=g).half()
    w = (torch.randn((cout, 9 * cin), device=cuda, generator=g) * (9 * cin) ** -0.5).half()
    bias = torch.randn((cout,), device=cuda, generator=g)
    scale = torch.rand((cout,), device=cuda, generator=g) + 0.5
    ref = K.conv3x3(x, w, bias, act=K.ACT_NONE, stride=stride, scale=scale, out_dtype=torch.float32)
    parts = K.conv3x3(x, w, bias, act=K.ACT_NONE, stride=stride, scale=scale, out_dtype=torch.float32, split_k=S)
    assert parts.shape[0] == S and parts.shape[1:] == ref.shape
    tot = parts[0].clone()
    for s in range(1, S):
        tot += parts[s]
    err = float((tot - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err
    assert float(parts[1:].abs().max()) > 0 and float((parts[0] - ref).abs().max()) > 1e-3, "the k range was not split"
    alone = K.conv3x3(x[n - 1:n].contiguous(), w, bias, act=K.ACT_NONE, stride=stride, scale=scale, out_dtype=torch.float32, split_k=S)
    assert torch.equal(alone[:, 0], parts[:, n - 1]), "a frame's partial sums depend on the batch"
    out3 = torch.empty(ref.shape[:3] + (3 * cout,), dtype=torch.float16, device=cuda)
    K.split3(parts, K.ACT_SILU, out3)
    v = tot / (1 + torch.exp(-tot))
    hi = out3[..., :cout].float()
    lo = out3[..., cout:2 * cout].float() / 2048
    assert torch.equal(out3[..., 2 * cout:], out3[..., :cout])
    assert float((hi + lo - v).abs().max()) <= 2e-6 * float(v.abs().max()) + 1e-7


@pytest.mark.parametrize("ln_inside", [True, False])
@pytest.mark.parametrize("n,Gh,Gw", [(1, 8, 8), (2, 16, 24), (3, 64, 64)])
def test_hiera_attn8_fused_block(cuda, n, Gh, Gw, ln_inside):
    """lmx_k_hiera_attn8 (csrc/hiera.hip): x += proj(window attention(qkv(layer_norm1(x)))) for 8 x 8-token windows in one launch,
    against (a) the fp32 definition (TF sam2 Sam2MultiScaleBlock / Sam2MultiScaleAttention over window_partition'ed tokens) with the
    f16-rounded weights and (b) the four launches it replaces (same rounding points; the f32 sums run in a different order); in both
    forms: layer_norm1 inside the kernel, or its f16 rows handed in."""
    from lmx import kernels as Kk
    from lmx import sam

    D, heads, hd, eps = 112, 2, 56, 1e-6
    rows = n * Gh * Gw
    x = _rand((rows, D), 72, 1.0) + 0.3
    gam, bet = 1.0 + _rand((D,), 77, 0.2), _rand((D,), 78, 0.2)
    wqkv = (_rand((3 * D, D), 73, 1.0) * D ** -0.5).half().float()
    bqkv = _rand((3 * D,), 74, 0.2)
    wo = (_rand((D, D), 75, 1.0) * D ** -0.5).half().float()
    bo = _rand((D,), 76, 0.2)
    # (a) fp32 definition
    h = torch.nn.functional.layer_norm(x, (D,), gam, bet, eps)
    qkv = h @ wqkv.t() + bqkv
    t = qkv.view(n, Gh // 8, 8, Gw // 8, 8, 3, heads, hd).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, -1, heads, 64, hd)
    att = torch.softmax(t[0] @ t[1].transpose(-1, -2) * hd ** -0.5, -1) @ t[2]  # [windows, heads, 64, hd]
    att = att.view(n, Gh // 8, Gw // 8, heads, 8, 8, hd).permute(0, 1, 4, 2, 5, 3, 6).reshape(rows, D)
    ref = x + att @ wo.t() + bo
    # device
    packed = tuple(torch.from_numpy(a).to(cuda) for a in sam.pack_hiera_attn(wqkv.numpy(), bqkv.numpy(), wo.numpy(), bo.numpy(), heads,
                                                                             ln_inside=ln_inside))
    xd = x.to(cuda)
    if ln_inside:
        Kk.hiera_attn8(xd, packed, n, Gh, Gw, heads, ln=(gam.to(cuda), bet.to(cuda), eps))
    else:
        Kk.hiera_attn8(xd, packed, n, Gh, Gw, heads, h=Kk.layernorm(xd, gam.to(cuda), bet.to(cuda), eps))
    _close(xd, ref, 4e-3, 4e-3, f"hiera_attn8 n{n} {Gh}x{Gw} vs fp32")
    # (b) the unfused launches
    xu = x.to(cuda)
    hd_ = Kk.layernorm(xu, gam.to(cuda), bet.to(cuda), eps)
    q3 = Kk.gemm(hd_, wqkv.half().to(cuda), bias=bqkv.to(cuda))
    a = torch.empty((rows, D), dtype=torch.float16, device=cuda)
    Kk.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], a, n * (Gh // 8) * (Gw // 8), heads, 64, 64, hd, hd ** -0.5,
                 window=dict(Gh=Gh, Gw=Gw, ws=8, q_stride=1), pad_k=q3[0, D:2 * D].contiguous(), pad_v=q3[0, 2 * D:].contiguous())
    Kk.gemm(a, wo.half().to(cuda), bias=bo.to(cuda), res=xu, out=xu)
    d = (xd - xu).abs().max().item()
    print(f"hiera_attn8 n{n} {Gh}x{Gw}: max |fused - unfused| {d:.3e}, max |fused - fp32| {(xd.cpu() - ref).abs().max().item():.3e}, "
          f"max |unfused - fp32| {(xu.cpu() - ref).abs().max().item():.3e}")
    assert d < 4e-3


@pytest.mark.parametrize("n,Gh,Gw", [(1, 4, 4), (1, 8, 12), (2, 32, 32), (3, 20, 28)])
def test_hiera_attn4_fused_block(cuda, n, Gh, Gw):
    """lmx_k_hiera_attn4 (csrc/hiera.hip): x += proj(window attention(qkv(h))) for 4 x 4-token windows at D = 224 (4 heads), weights
    streamed through LDS; against the fp32 definition and the three launches it replaces.  The grids cover a partial last group of
    16 windows, a single window, and several groups per workgroup."""
    from lmx import kernels as Kk
    from lmx import sam

    D, heads, hd = 224, 4, 56
    rows = n * Gh * Gw
    h = _rand((rows, D), 81, 1.0).half()
    x = _rand((rows, D), 82, 1.0)
    wqkv = (_rand((3 * D, D), 83, 1.0) * D ** -0.5).half().float()
    bqkv = _rand((3 * D,), 84, 0.2)
    wo = (_rand((D, D), 85, 1.0) * D ** -0.5).half().float()
    bo = _rand((D,), 86, 0.2)
    qkv = h.float() @ wqkv.t() + bqkv
    t = qkv.view(n, Gh // 4, 4, Gw // 4, 4, 3, heads, hd).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, -1, heads, 16, hd)
    att = torch.softmax(t[0] @ t[1].transpose(-1, -2) * hd ** -0.5, -1) @ t[2]
    att = att.view(n, Gh // 4, Gw // 4, heads, 4, 4, hd).permute(0, 1, 4, 2, 5, 3, 6).reshape(rows, D)
    ref = x + att @ wo.t() + bo
    packed = tuple(torch.from_numpy(a).to(cuda) for a in sam.pack_hiera_attn4(wqkv.numpy(), bqkv.numpy(), wo.numpy(), bo.numpy(), heads))
    xd = x.to(cuda)
    Kk.hiera_attn4(h.to(cuda), xd, packed, n, Gh, Gw, heads)
    _close(xd, ref, 4e-3, 4e-3, f"hiera_attn4 n{n} {Gh}x{Gw} vs fp32")
    q3 = Kk.gemm(h.to(cuda), wqkv.half().to(cuda), bias=bqkv.to(cuda))
    a = torch.empty((rows, D), dtype=torch.float16, device=cuda)
    Kk.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], a, n * (Gh // 4) * (Gw // 4), heads, 16, 16, hd, hd ** -0.5,
                 window=dict(Gh=Gh, Gw=Gw, ws=4, q_stride=1), pad_k=q3[0, D:2 * D].contiguous(), pad_v=q3[0, 2 * D:].contiguous())
    xu = x.to(cuda)
    Kk.gemm(a, wo.half().to(cuda), bias=bo.to(cuda), res=xu, out=xu)
    d = (xd - xu).abs().max().item()
    print(f"hiera_attn4 n{n} {Gh}x{Gw}: max |fused - unfused| {d:.3e}, max |fused - fp32| {(xd.cpu() - ref).abs().max().item():.3e}, "
          f"max |unfused - fp32| {(xu.cpu() - ref).abs().max().item():.3e}")
    assert d < 4e-3


@pytest.mark.parametrize("Din,D,heads,ws,n,Gh,Gw", [(112, 224, 4, 8, 1, 8, 8), (112, 224, 4, 8, 1, 16, 24), (112, 224, 4, 8, 2, 64, 64),
                                                      (112, 224, 4, 8, 3, 24, 40), (224, 448, 8, 4, 1, 4, 4), (224, 448, 8, 4, 1, 8, 12),
                                                      (224, 448, 8, 4, 2, 32, 32), (224, 448, 8, 4, 3, 20, 28)])
def test_hiera_attn_pool_fused_block(cuda, Din, D, heads, ws, n, Gh, Gw):
    """lmx_k_hiera_attn_pool (csrc/hiera.hip): the attention half of the blocks that open Hiera stages 2 and 3 — pool(proj(h)) +
    attn_proj(window attention(pool(q), k, v)): 112 -> 224 channels on 8 x 8 windows, 224 -> 448 on 4 x 4 windows, 2 x 2 max-pooled
    queries and shortcut — against the fp32 definition (TF sam2 Sam2MultiScaleBlock / Sam2MultiScaleAttention with q_stride) and the
    five launches it replaces.  The grids cover single windows, partial last groups and several groups per workgroup."""
    from lmx import kernels as Kk
    from lmx import sam

    hd, wq = D // heads, ws // 2
    rows = n * Gh * Gw
    h = _rand((rows, Din), 91, 1.0).half()
    wsc = (_rand((D, Din), 92, 1.0) * Din ** -0.5).half().float()
    bsc = _rand((D,), 93, 0.2)
    wqkv = (_rand((3 * D, Din), 94, 1.0) * Din ** -0.5).half().float()
    bqkv = _rand((3 * D,), 95, 0.2)
    wo = (_rand((D, D), 96, 1.0) * D ** -0.5).half().float()
    bo = _rand((D,), 97, 0.2)

    def pool(t):  # [n, Gh, Gw, C] -> 2 x 2 max pool
        return torch.nn.functional.max_pool2d(t.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)

    hf = h.float()
    sc = pool((hf @ wsc.t() + bsc).view(n, Gh, Gw, D))
    qkv = (hf @ wqkv.t() + bqkv).view(n, Gh, Gw, 3, D)
    q = pool(qkv[..., 0, :])  # [n, Gh/2, Gw/2, D]
    q = q.reshape(n, Gh // ws, wq, Gw // ws, wq, heads, hd).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, heads, wq * wq, hd)
    kk, vv = (qkv[..., j, :].reshape(n, Gh // ws, ws, Gw // ws, ws, heads, hd).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, heads, ws * ws, hd) for j in (1, 2))
    att = torch.softmax(q @ kk.transpose(-1, -2) * hd ** -0.5, -1) @ vv  # [windows, heads, wq * wq, hd]
    att = att.view(n, Gh // ws, Gw // ws, heads, wq, wq, hd).permute(0, 1, 4, 2, 5, 3, 6).reshape(n, Gh // 2, Gw // 2, D)
    ref = (sc + att @ wo.t() + bo).reshape(-1, D)
    packed = tuple(torch.from_numpy(a).to(cuda) for a in sam.pack_hiera_attn_pool(wsc.numpy(), bsc.numpy(), wqkv.numpy(), bqkv.numpy(), wo.numpy(),
                                                                                  bo.numpy(), heads))
    out = Kk.hiera_attn_pool(h.to(cuda), packed, n, Gh, Gw, heads, D)
    _close(out, ref, 4e-3, 4e-3, f"hiera_attn_pool {Din}->{D} n{n} {Gh}x{Gw} vs fp32")
    # the unfused launches (as lmx/sam.py HieraEncoder._attention_half runs them)
    hd_ = h.to(cuda)
    scu = Kk.gemm(hd_, wsc.half().to(cuda), bias=bsc.to(cuda), out_dtype=torch.float32)
    pooled = torch.empty((n, Gh // 2, Gw // 2, D), dtype=torch.float32, device=cuda)
    Kk.maxpool2(scu.view(n, Gh, Gw, D), pooled)
    q3 = Kk.gemm(hd_, wqkv.half().to(cuda), bias=bqkv.to(cuda))
    qp = torch.empty((n, Gh // 2, Gw // 2, D), dtype=torch.float16, device=cuda)
    Kk.maxpool2(q3.view(n, Gh, Gw, 3 * D)[..., :D], qp)
    a = torch.empty((rows // 4, D), dtype=torch.float16, device=cuda)
    Kk.attention(qp.view(-1, D), q3[:, D:2 * D], q3[:, 2 * D:], a, n * (Gh // ws) * (Gw // ws), heads, wq * wq, ws * ws, hd, hd ** -0.5,
                 window=dict(Gh=Gh, Gw=Gw, ws=ws, q_stride=2), pad_k=q3[0, D:2 * D].contiguous(), pad_v=q3[0, 2 * D:].contiguous())
    xu = torch.empty((rows // 4, D), dtype=torch.float32, device=cuda)
    Kk.gemm(a, wo.half().to(cuda), bias=bo.to(cuda), res=pooled.view(-1, D), out=xu)
    d = (out - xu).abs().max().item()
    print(f"hiera_attn_pool {Din}->{D} n{n} {Gh}x{Gw}: max |fused - unfused| {d:.3e}, max |fused - fp32| {(out.cpu() - ref).abs().max().item():.3e}, "
          f"max |unfused - fp32| {(xu.cpu() - ref).abs().max().item():.3e}")
    assert d < 4e-3


@pytest.mark.parametrize("D,rows", [(112, 256), (112, 5000), (224, 300), (224, 70000)])
def test_ln_mlp_img(cuda, D, rows):
    """lmx_k_ln_mlp_img (csrc/hiera.hip hiera_mlp_kernel): x += fc2(gelu(fc1(LayerNorm(x)))) with the weights streamed as LDS images,
    with the f16 copy and the next block's LayerNorm rows, against the fp32 definition and against lmx_k_ln_mlp (csrc/mlp.hip: same
    rounding points).  Row counts cover a partial last group of 256 tokens and several groups per workgroup."""
    from lmx import kernels as Kk
    from lmx import sam

    x = _rand((rows, D), 101, 1.0) + 0.2
    g2, e2 = 1.0 + _rand((D,), 102, 0.2), _rand((D,), 103, 0.2)
    gn, en = 1.0 + _rand((D,), 104, 0.2), _rand((D,), 105, 0.2)
    w1 = (_rand((4 * D, D), 106, 1.0) * D ** -0.5).half().float()
    b1 = _rand((4 * D,), 107, 0.2)
    w2 = (_rand((D, 4 * D), 108, 1.0) * (4 * D) ** -0.5).half().float()
    b2 = _rand((D,), 109, 0.2)
    eps = 1e-6
    hh = torch.nn.functional.layer_norm(x, (D,), g2, e2, eps)
    ref = x + torch.nn.functional.gelu(hh @ w1.t() + b1) @ w2.t() + b2
    refn = torch.nn.functional.layer_norm(ref, (D,), gn, en, eps)
    packed = tuple(torch.from_numpy(a).to(cuda) for a in sam.pack_ln_mlp(w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), g2.numpy(), e2.numpy(),
                                                                         gn.numpy(), en.numpy()))
    xd = x.to(cuda)
    x16 = torch.zeros((rows, D), dtype=torch.float16, device=cuda)
    hn = torch.zeros((rows, D), dtype=torch.float16, device=cuda)
    Kk.ln_mlp_img(xd, packed, eps, x16=x16, h_next=hn)
    _close(xd, ref, 4e-3, 4e-3, f"ln_mlp_img D{D} rows{rows} vs fp32")
    _close(hn, refn, 6e-3, 6e-3, f"ln_mlp_img D{D} rows{rows} h_next vs fp32")
    assert torch.equal(x16, xd.half())
    xu = x.to(cuda)
    x16u = torch.zeros_like(x16)
    hnu = torch.zeros_like(hn)
    Kk.ln_mlp(xu, g2.to(cuda), e2.to(cuda), w1.half().to(cuda), b1.to(cuda), w2.half().to(cuda), b2.to(cuda), eps, x16=x16u,
              next_ln=(gn.to(cuda), en.to(cuda), hnu))
    d = (xd - xu).abs().max().item()
    print(f"ln_mlp_img D{D} rows{rows}: max |img - mlp.hip| {d:.3e}, max |img - fp32| {(xd.cpu() - ref).abs().max().item():.3e}, "
          f"max |mlp.hip - fp32| {(xu.cpu() - ref).abs().max().item():.3e}")
    assert d < 4e-3
