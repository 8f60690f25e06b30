"""Regression tests for the round-1 wrong-value defect.  Cause: v_pk_add_f32 / v_pk_mul_f32 with
op_sel taking src1's HIGH register misread their operand (up to 0.4 % of evaluations) while another wave of the same SIMD
issues MFMAs; hipcc's SLP vectoriser emitted that form in mask_post (plain-load build) and in the first fused MLP.  The
build now forbids the form (tools/isa_lint.py, -fno-slp-vectorize); these tests run the victims beside the aggressor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _beside_attention(cuda, foreground, n_iter=40):
    """Run `foreground()` (returns a 0-d count of wrong elements) n_iter times on one stream while three other streams
    run the 128-query-tile attention kernel (the MFMA-dense, 2-3 waves/SIMD kernel beside which the defect showed in
    148-150 of 150 launches)."""
    from lmx import kernels as K

    g = torch.Generator(device=cuda).manual_seed(1)
    qkv = torch.randn(16 * 4096, 3 * 448, device=cuda, generator=g).half()
    ao = torch.empty(16 * 4096, 448, device=cuda, dtype=torch.float16)
    main = torch.cuda.current_stream()
    fg = torch.cuda.Stream()
    bgs = [torch.cuda.Stream() for _ in range(3)]
    bad = torch.zeros((), dtype=torch.int64, device=cuda)
    for st in [fg] + bgs:
        st.wait_stream(main)
    for _ in range(n_iter):
        for st in bgs:
            with torch.cuda.stream(st):
                K.attention(qkv[:, :448], qkv[:, 448:896], qkv[:, 896:], ao, 16, 7, 4096, 4096, 64, 0.125)
        with torch.cuda.stream(fg):
            bad += foreground()
    torch.cuda.synchronize()
    return int(bad)


def test_mask_post_beside_mfma_attention(cuda):
    from lmx import kernels as K

    g = torch.Generator(device=cuda).manual_seed(1)
    logits = torch.randn(16, 256, 256, device=cuda, generator=g)
    ref, ref_stats = K.mask_post(logits, 1024, 576, 1024, 1080, 1920)
    torch.cuda.synchronize()

    def fgd():
        m, st = K.mask_post(logits, 1024, 576, 1024, 1080, 1920)
        return (m != ref).sum() + (st != ref_stats).sum()

    assert _beside_attention(cuda, fgd) == 0


def test_fused_mlp_and_layernorm_beside_mfma_attention(cuda):
    """The other kernel family that showed the signature (row statistics of the first fused MLP): LayerNorm + fused MLP of
    a Hiera stage-1 shape, bit-identical to the idle result while MFMA attention runs on three other streams."""
    from lmx import kernels as K

    g = torch.Generator(device=cuda).manual_seed(2)
    D, rows = 112, 262144
    x = torch.randn(rows, D, device=cuda, generator=g)
    gam, bet = torch.randn(D, device=cuda, generator=g), torch.randn(D, device=cuda, generator=g)
    w1 = (torch.randn(4 * D, D, device=cuda, generator=g) * 0.1).half()
    w2 = (torch.randn(D, 4 * D, device=cuda, generator=g) * 0.05).half()
    b1, b2 = torch.randn(4 * D, device=cuda, generator=g) * 0.1, torch.randn(D, device=cuda, generator=g) * 0.1

    def run():
        xx = x.clone()
        K.ln_mlp(xx, gam, bet, w1, b1, w2, b2, 1e-6)
        return xx

    ref = run()
    torch.cuda.synchronize()
    assert _beside_attention(cuda, lambda: (run() != ref).sum(), n_iter=25) == 0


def test_fused_attention_kernels_reproducible_beside_mfma_kernels(cuda):
    """Round-3 regression: the softmax row maximum of the fused Hiera attention kernels once ran as inline-asm
    v_max3_f32 on MFMA results — hipcc inserts the MFMA-write -> VALU-read wait states only in front of instructions it knows, so
    the asm read accumulators that were not written yet whenever a second wave kept the matrix pipe busy: a different (still
    valid) softmax shift from run to run, invisible to every tolerance.  Each kernel, repeated beside a GEMM stream, must return
    the same bits every time (tools/hiera_pool_determinism.py is the stand-alone form: 19 of 30 runs differed before the fix)."""
    import numpy as np
    import torch

    from lmx import kernels as K
    from lmx import sam

    g = torch.Generator().manual_seed(11)

    def rnd(*shape, scale=1.0):
        return torch.randn(shape, generator=g) * scale

    n, G = 2, 256
    cases = []
    # stage 1 (8 x 8 windows, D = 112), both forms
    D, heads = 112, 2
    wqkv, bqkv, wo, bo = rnd(3 * D, D, scale=D ** -0.5).half().float(), rnd(3 * D, scale=0.2), rnd(D, D, scale=D ** -0.5).half().float(), rnd(D, scale=0.2)
    x1 = rnd(n * G * G, D).to(cuda)
    gam, bet = torch.ones(D, device=cuda), torch.zeros(D, device=cuda)
    for ln_inside in (True, False):
        pk = tuple(torch.from_numpy(a).to(cuda) for a in sam.pack_hiera_attn(wqkv.numpy(), bqkv.numpy(), wo.numpy(), bo.numpy(), heads, ln_inside=ln_inside))
        h1 = K.layernorm(x1, gam, bet, 1e-6)
        if ln_inside:
            cases.append(("hiera_attn8/ln", lambda pk=pk: K.hiera_attn8(x1.clone(), pk, n, G, G, heads, ln=(gam, bet, 1e-6))))
        else:
            cases.append(("hiera_attn8", lambda pk=pk, h1=h1: K.hiera_attn8(x1.clone(), pk, n, G, G, heads, h=h1)))
    # the block that opens stage 2 (112 -> 224, pooled), and stage 2 (4 x 4 windows, D = 224)
    Do, heads4 = 224, 4
    wsc, bsc = rnd(Do, D, scale=D ** -0.5).half().float(), rnd(Do, scale=0.2)
    wq2, bq2 = rnd(3 * Do, D, scale=D ** -0.5).half().float(), rnd(3 * Do, scale=0.2)
    wo2, bo2 = rnd(Do, Do, scale=Do ** -0.5).half().float(), rnd(Do, scale=0.2)
    pkp = tuple(torch.from_numpy(a).to(cuda) for a in sam.pack_hiera_attn_pool(wsc.numpy(), bsc.numpy(), wq2.numpy(), bq2.numpy(), wo2.numpy(), bo2.numpy(), heads4))
    hp = rnd(n * G * G, D).half().to(cuda)
    cases.append(("hiera_attn_pool", lambda: K.hiera_attn_pool(hp, pkp, n, G, G, heads4, Do)))
    wq4, bq4 = rnd(3 * Do, Do, scale=Do ** -0.5).half().float(), rnd(3 * Do, scale=0.2)
    pk4 = tuple(torch.from_numpy(a).to(cuda) for a in sam.pack_hiera_attn4(wq4.numpy(), bq4.numpy(), wo2.numpy(), bo2.numpy(), heads4))
    G4 = G // 2
    h4, x4 = rnd(n * G4 * G4, Do).half().to(cuda), rnd(n * G4 * G4, Do).to(cuda)
    cases.append(("hiera_attn4", lambda: K.hiera_attn4(h4, x4.clone(), pk4, n, G4, G4, heads4)))

    a = torch.randn((65536, 448), device=cuda).half()
    w = torch.randn((1792, 448), device=cuda).half()
    side = torch.cuda.Stream()
    for name, run in cases:
        ref = run().clone()
        torch.cuda.synchronize()
        for it in range(12):
            with torch.cuda.stream(side):
                for _ in range(3):
                    K.gemm(a, w, act=K.ACT_GELU)
            out = run()
            torch.cuda.synchronize()
            assert torch.equal(out, ref), f"{name}: run {it} differs from the first in {int((out != ref).sum())} elements"
    assert np.isfinite(ref.float().cpu().numpy()).all()
