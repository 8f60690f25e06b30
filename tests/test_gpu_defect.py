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
