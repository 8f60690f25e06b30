"""Thin torch-tensor front end of the C-ABI (include/lmx.h).  PyTorch is used for device memory and streams only:
every function below checks shapes/strides on the host, fills the C descriptor and enqueues ONE liblmx kernel on
torch's current HIP stream.  No arithmetic happens in Python and there is no fallback path."""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import AttnDesc, GemmDesc, LmxError, check

F16, F32 = 0, 1
ACT_NONE, ACT_SILU, ACT_GELU, ACT_RELU = 0, 1, 2, 3
_DT = {torch.float16: F16, torch.float32: F32}


_tls = threading.local()  # .dev: device of the launch being issued by THIS thread (read by the launch trace only)


def _stream(dev):
    """The HIP stream the launch goes to: torch's current stream OF THE OPERANDS' DEVICE `dev` (what _dev returned), not of
    the current device.  The device is passed explicitly: two threads driving two devices do not share any state here."""
    _tls.dev = dev
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev(*ts):
    """Every operand must live in HBM, all on one device; returns that device (the argument of _stream)."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise LmxError("lmx kernels take device (HBM) tensors only; got a CPU tensor")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise LmxError(f"lmx kernels: operands on different devices ({dev} and {t.device})")
    if dev is None:
        raise LmxError("lmx kernels: no device operand")
    return dev


def _rows(t, what):
    """2-D row-major view: last dim contiguous; returns (rows, cols, row_stride)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise LmxError(f"{what}: expected a 2-D tensor with contiguous last dim, got shape {tuple(t.shape)} "
                       f"strides {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def pooled_gemm_ok(M, N):
    """Shapes lmx_k_gemm's pooled-row mode (a_mode 2) is built for (the LDS-DMA kernel); smaller ones take GEMM + maxpool2."""
    return M >= 512 and N >= 96 and N % 8 == 0


def gemm(a, w, bias=None, act=ACT_NONE, scale=None, res=None, out=None, out_dtype=torch.float16, res_rows=0, pool_hw=None, a_rep=1):
    """out[M,N] = res + scale * act(a[M,K] @ w[N,K]^T + bias)   (lmx_k_gemm, a_mode 0).
    a_rep > 1: w is [N, a_rep*K] and a's K columns are walked a_rep times (w = [whi | wlo]: f16 activations against 22-bit
    weights in one launch).
    res_rows > 0: res is a [res_rows, N] table broadcast over the batch (row m uses res[m % res_rows]).
    pool_hw=(H, W): the rows of `a` are an [n, H, W] token grid and the result is the 2 x 2 max-pool of the product over that
    grid, [M/4, N] in [n, H/2, W/2] order (a_mode 2: the bits of gemm(...) followed by maxpool2, without the full-size
    intermediate)."""
    dev = _dev(a, w, bias, scale, res, out)
    M, K, lda = _rows(a, "gemm A")
    N, K2, ldw = _rows(w, "gemm W")
    if K2 != a_rep * K or ldw != K2:
        raise LmxError(f"gemm: W must be contiguous [N,{a_rep}*K]; A K={K}, W shape {tuple(w.shape)}")
    if a.dtype != torch.float16 or w.dtype != torch.float16:
        raise LmxError("gemm: A and W must be float16")
    Mout = M // 4 if pool_hw else M
    if out is None:
        out = torch.empty((Mout, N), dtype=out_dtype, device=a.device)
    Mo, No, ldc = _rows(out, "gemm C")
    if (Mo, No) != (Mout, N):
        raise LmxError(f"gemm: out shape {tuple(out.shape)} != ({Mout},{N})")
    d = GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.scale = scale.data_ptr() if scale is not None else None
    d.lda, d.ldc, d.ldr = lda, ldc, 0
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
        raise LmxError("gemm: bias must be float32 [N]")
    if scale is not None and (scale.dtype != torch.float32 or scale.numel() != N):
        raise LmxError("gemm: scale must be float32 [N]")
    if res is not None:
        Mr, Nr, ldr = _rows(res, "gemm res")
        if (Mr, Nr) != ((res_rows or M), N) or res.dtype != out.dtype:
            raise LmxError("gemm: residual must match out's shape and dtype")
        d.res, d.ldr = res.data_ptr(), ldr
    d.res_rows = res_rows
    d.M, d.N, d.K = M, N, a_rep * K
    d.act, d.out_dtype, d.a_mode, d.a_rep = act, _DT[out.dtype], 0, a_rep
    if pool_hw:
        d.a_mode, d.H, d.W_ = 2, int(pool_hw[0]), int(pool_hw[1])
    check(_lib.load().lmx_k_gemm(C.byref(d), _stream(dev)), "lmx_k_gemm")
    return out


def im2col_u8(img, lut, IH, IW, KH, KW, stride, pad, ldo):
    """u8 [n,rh,rw,3] at the top-left of an IH x IW zero canvas -> f16 [n*OH*OW, ldo] (lmx_k_im2col_u8)."""
    dev = _dev(img, lut)
    n, rh, rw, c = img.shape
    if c != 3 or img.dtype != torch.uint8 or not img.is_contiguous():
        raise LmxError("im2col_u8: img must be contiguous uint8 [n,h,w,3]")
    OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
    out = torch.empty((n * OH * OW, ldo), dtype=torch.float16, device=img.device)
    check(_lib.load().lmx_k_im2col_u8(_ptr(img), _ptr(lut), _ptr(out), n, rh, rw, IH, IW, KH, KW, stride, pad, ldo,
                                      _stream(dev)), "lmx_k_im2col_u8")
    return out


def maxpool2(x, out):
    """2x2/s2 max pool, NHWC f16 or f32, channel slices allowed (lmx_k_maxpool2)."""
    dev = _dev(x, out)
    n, H, W, Cc, ps = _nhwc(x, "maxpool2 x")
    no, Ho, Wo, Co, pso = _nhwc(out, "maxpool2 out")
    if (no, Ho, Wo, Co) != (n, H // 2, W // 2, Cc) or x.dtype != out.dtype:
        raise LmxError("maxpool2: shape/dtype mismatch")
    check(_lib.load().lmx_k_maxpool2(_ptr(x), ps, _ptr(out), pso, _DT[x.dtype], n, H, W, Cc, _stream(dev)), "lmx_k_maxpool2")
    return out


def cast_f16(x, out=None):
    dev = _dev(x, out)
    rows, cols, lds = _rows(x, "cast src")
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float16, device=x.device)
    check(_lib.load().lmx_k_cast_f32_f16(_ptr(x), lds, _ptr(out), out.stride(0), rows, cols, _stream(dev)),
          "lmx_k_cast_f32_f16")
    return out


def _nhwc(t, what):
    """[n,H,W,C] view of (a channel slice of) a dense NHWC buffer; returns (n,H,W,C,pixel_stride)."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise LmxError(f"{what}: expected NHWC with contiguous channels, got {tuple(t.shape)} / {t.stride()}")
    n, H, W, Cc = t.shape
    ps = t.stride(2)
    if t.stride(1) != W * ps or (n > 1 and t.stride(0) != H * W * ps):
        raise LmxError(f"{what}: not a channel slice of a dense NHWC buffer: strides {t.stride()}")
    return n, H, W, Cc, ps


def split_k_for(px_per_frame, N, K, cin):
    """Split factor of an f32-output 3 x 3 convolution in the exact plan (lmx_k_gemm split_k): K = 27 Cin against a handful of
    256 x 256 tiles per frame leaves most of the 256 CUs idle at the reference schedule's 10 frames (M = 9600, N = 256, K = 6912:
    38 tiles, 122 us).  The factor is a function of the LAYER — pixels of ONE frame, N, K — never of the batch: the summation
    order, and so a frame's bits, must not depend on the batch the frame rides in.  1 = no split (also outside the LDS-DMA
    kernel's shapes)."""
    if N < 64 or N % 8 or cin % 32:
        return 1
    tiles1 = -(-px_per_frame // 256) * -(-N // 256)  # tiles one frame contributes
    nk = K // 64
    s = min(8, nk // 8, 32 // tiles1)
    return s if s >= 2 else 1


def conv3x3(x, w, bias=None, act=ACT_SILU, stride=1, res=None, out=None, scale=None, out_dtype=torch.float16, split_k=1):
    """3x3 / pad 1 convolution as implicit GEMM (lmx_k_gemm, a_mode 1).  x: NHWC f16 (may be a channel slice),
    w: f16 [Cout, 9*Cin] packed (ky,kx,ci); out: NHWC f16 or f32 (may be a channel slice of a wider buffer);
    scale: f32 [Cout] applied after bias + activation (the exact plan's power-of-two row scales)."""
    dev = _dev(x, w, bias, res, out, scale)
    n, H, W, Cin, ps = _nhwc(x, "conv3x3 x")
    Cout = w.shape[0]
    if w.dim() != 2 or w.shape[1] != 9 * Cin or not w.is_contiguous():
        raise LmxError(f"conv3x3: W must be contiguous [Cout, 9*Cin={9 * Cin}], got {tuple(w.shape)}")
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    parts = None
    if split_k > 1:  # S partial outputs [S, n, Ho, Wo, Cout] f32; the consumer (split3 nsum=S) adds them up
        if out is not None or res is not None or act != ACT_NONE or out_dtype != torch.float32:
            raise LmxError("conv3x3: split_k needs a fresh f32 output, no residual, no activation")
        parts = torch.empty((split_k, n, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        out = parts[0]
    if out is None:
        out = torch.empty((n, Ho, Wo, Cout), dtype=out_dtype, device=x.device)
    no, Ho2, Wo2, Co2, pso = _nhwc(out, "conv3x3 out")
    if (no, Ho2, Wo2, Co2) != (n, Ho, Wo, Cout):
        raise LmxError(f"conv3x3: out shape {tuple(out.shape)} != {(n, Ho, Wo, Cout)}")
    d = GemmDesc()
    d.A, d.W, d.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    if scale is not None and (scale.dtype != torch.float32 or scale.numel() != Cout):
        raise LmxError("conv3x3: scale must be float32 [Cout]")
    d.scale = scale.data_ptr() if scale is not None else None
    d.lda, d.ldc, d.ldr = ps, pso, 0
    if res is not None:
        nr, Hr, Wr, Cr, psr = _nhwc(res, "conv3x3 res")
        if (nr, Hr, Wr, Cr) != (n, Ho, Wo, Cout) or res.dtype != out.dtype:
            raise LmxError("conv3x3: residual must match out")
        d.res, d.ldr = res.data_ptr(), psr
    d.M, d.N, d.K = n * Ho * Wo, Cout, 9 * Cin
    d.act, d.out_dtype, d.a_mode = act, _DT[out.dtype], 1
    d.H, d.W_, d.Cin, d.conv_stride, d.Ho, d.Wo = H, W, Cin, stride, Ho, Wo
    if parts is not None:
        d.split_k, d.split_stride = split_k, parts.stride(0)
    check(_lib.load().lmx_k_gemm(C.byref(d), _stream(dev)), "lmx_k_gemm(conv3x3)")
    return out if parts is None else parts


def conv1x1(x, w, bias=None, act=ACT_SILU, res=None, out=None, out_dtype=torch.float16, scale=None):
    """1x1 convolution = GEMM over the pixels of an NHWC tensor (channel slices allowed on both sides)."""
    n, H, W, Cin, ps = _nhwc(x, "conv1x1 x")
    a = x.as_strided((n * H * W, Cin), (ps, 1))
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((n, H, W, Cout), dtype=out_dtype, device=x.device)
    no, Ho, Wo, Co, pso = _nhwc(out, "conv1x1 out")
    o2 = out.as_strided((n * H * W, Cout), (pso, 1))
    r2 = None
    if res is not None:
        nr, Hr, Wr, Cr, psr = _nhwc(res, "conv1x1 res")
        r2 = res.as_strided((n * H * W, Cout), (psr, 1))
    gemm(a, w, bias=bias, act=act, res=r2, out=o2, scale=scale)
    return out


def layernorm(x, gamma, beta, eps, out=None, out_dtype=torch.float16, act=ACT_NONE):
    dev = _dev(x, gamma, beta, out)
    rows, D, ldx = _rows(x, "layernorm x")
    if out is None:
        out = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    r2, D2, ldy = _rows(out, "layernorm y")
    if (r2, D2) != (rows, D):
        raise LmxError("layernorm: out shape mismatch")
    check(_lib.load().lmx_k_layernorm(_ptr(x), _DT[x.dtype], ldx, _ptr(gamma), _ptr(beta), _ptr(out), _DT[out.dtype],
                                      ldy, rows, D, float(eps), act, _stream(dev)), "lmx_k_layernorm")
    return out


def pack_bits(mask):
    """u8 [..., h, w] (0 / non-0) -> u8 [..., h, ceil(w/8)], numpy.packbits bit order."""
    dev = _dev(mask)
    if mask.dtype not in (torch.uint8, torch.bool) or not mask.is_contiguous():
        raise LmxError("pack_bits: contiguous uint8 / bool tensor expected")
    w = mask.shape[-1]
    rows = mask.numel() // w
    out = torch.empty(tuple(mask.shape[:-1]) + ((w + 7) // 8,), dtype=torch.uint8, device=mask.device)
    check(_lib.load().lmx_k_pack_bits(_ptr(mask), rows, w, _ptr(out), _stream(dev)), "lmx_k_pack_bits")
    return out


FUSED_MLP_WIDTHS = (112, 224)  # (448 exists as a development configuration: no faster than the unfused launches, csrc/mlp.hip)
if os.environ.get("LMX_MLP448"):  # development A/B only
    FUSED_MLP_WIDTHS = (112, 224, 448)


def ln_mlp(x, gamma, beta, w1, b1, w2, b2, eps, x16=None, next_ln=None):
    """x (f32 [rows, D], updated in place) += fc2(gelu(fc1(LayerNorm(x)))) for D in FUSED_MLP_WIDTHS (csrc/mlp.hip).
    x16: optional f16 [rows, D] that also receives the updated x (saves a cast pass where an f16 copy is needed next).
    next_ln=(gamma, beta, h): h (f16 [rows, D]) receives LayerNorm(updated x; gamma, beta, eps) — the next block's first
    LayerNorm, without its launch."""
    dev = _dev(x, gamma, beta, w1, b1, w2, b2, x16, *(next_ln or ()))
    rows, D, ldx = _rows(x, "ln_mlp x")
    if x.dtype != torch.float32 or w1.dtype != torch.float16 or w2.dtype != torch.float16:
        raise LmxError("ln_mlp: x must be f32 and the weights f16")
    if tuple(w1.shape) != (4 * D, D) or tuple(w2.shape) != (D, 4 * D) or not (w1.is_contiguous() and w2.is_contiguous()):
        raise LmxError("ln_mlp: weight shapes must be [4D, D] and [D, 4D], contiguous")
    if x16 is not None and (x16.dtype != torch.float16 or tuple(x16.shape) != (rows, D) or not x16.is_contiguous()):
        raise LmxError("ln_mlp: x16 must be contiguous float16 [rows, D]")
    gn, bn, hn = next_ln if next_ln else (None, None, None)
    if hn is not None and (hn.dtype != torch.float16 or tuple(hn.shape) != (rows, D) or not hn.is_contiguous() or gn.numel() != D or bn.numel() != D):
        raise LmxError("ln_mlp: next_ln = (gamma [D], beta [D], contiguous float16 [rows, D])")
    ws = torch.empty((rows, D), dtype=torch.float16, device=x.device)  # LayerNorm output (the library's workspace)
    check(_lib.load().lmx_k_ln_mlp(_ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), rows, D,
                                   float(eps), _ptr(ws), _ptr(x16), _ptr(gn), _ptr(bn), _ptr(hn), _stream(dev)), "lmx_k_ln_mlp")
    return x


def hiera_attn8_ok(D, heads, win, Gh, Gw, q_stride):
    """Shapes lmx_k_hiera_attn8 is built for: Hiera-B+ stage 1 (D 112, 2 heads of 56), whole 8 x 8 windows, no query pooling."""
    return D == 112 and heads == 2 and win == 8 and not q_stride and Gh % 8 == 0 and Gw % 8 == 0 and os.environ.get("LMX_HIERA_ATTN8", "1") != "0"


def hiera_attn8(x, packed, n_img, Gh, Gw, heads, h=None, ln=None):
    """x (f32 [rows, D], in place) += proj(window attention(qkv(layer_norm1(x)))) for 8 x 8-token windows (csrc/hiera.hip).
    Either h = layer_norm1(x) as f16 [rows, D] (packed from pack_hiera_attn(..., ln_inside=False)) or ln = (gamma, beta, eps): the
    kernel normalises x itself (packed with ln_inside=True).  packed = (wqkv_p, bqkv_p, wo_p, bo)."""
    wq, bq, wo, bo = packed
    if (h is None) == (ln is None):
        raise LmxError("hiera_attn8: give either h or ln=(gamma, beta, eps)")
    gamma, beta, eps = ln if ln is not None else (None, None, 0.0)
    dev = _dev(x, h, gamma, beta, wq, bq, wo, bo)
    rows, D, ldx = _rows(x, "hiera_attn8 x")
    if x.dtype != torch.float32 or rows != n_img * Gh * Gw:
        raise LmxError("hiera_attn8: x must be float32 [n*Gh*Gw, D]")
    if h is not None and (h.dtype != torch.float16 or tuple(h.shape) != (rows, D) or not h.is_contiguous()):
        raise LmxError("hiera_attn8: h must be contiguous float16 [rows, D]")
    if ln is not None and (gamma.numel() != D or beta.numel() != D or gamma.dtype != torch.float32 or beta.dtype != torch.float32):
        raise LmxError("hiera_attn8: gamma / beta must be float32 [D]")
    if tuple(wq.shape) != (3 * heads * 64, 128) or tuple(wo.shape) != (D, heads * 64) or wq.dtype != torch.float16 or wo.dtype != torch.float16 \
            or not (wq.is_contiguous() and wo.is_contiguous()) or bq.numel() != 3 * heads * 64 or bo.numel() != D:
        raise LmxError("hiera_attn8: packed operands have the wrong shapes (lmx.sam.pack_hiera_attn)")
    check(_lib.load().lmx_k_hiera_attn8(_ptr(h), _ptr(x), ldx, _ptr(gamma), _ptr(beta), float(eps), _ptr(wq), _ptr(bq), _ptr(wo), _ptr(bo),
                                        n_img, Gh, Gw, D, heads, float((D // heads) ** -0.5), _stream(dev)), "lmx_k_hiera_attn8")
    return x


def hiera_attn4_ok(D, heads, win, Gh, Gw, q_stride):
    """Shapes lmx_k_hiera_attn4 is built for: Hiera-B+ stage 2 (D 224, 4 heads of 56), whole 4 x 4 windows, no query pooling."""
    return D == 224 and heads == 4 and win == 4 and not q_stride and Gh % 4 == 0 and Gw % 4 == 0 and os.environ.get("LMX_HIERA_ATTN4", "1") != "0"


def hiera_attn4(h, x, packed, n_img, Gh, Gw, heads):
    """x (f32 [rows, D], in place) += proj(window attention(qkv(h))) for 4 x 4-token windows, weights streamed (csrc/hiera.hip).
    h f16 [rows, D] contiguous = layer_norm1(x); packed = (w_img, bias) from lmx.sam.pack_hiera_attn4."""
    img, bias = packed
    dev = _dev(h, x, img, bias)
    rows, D, ldx = _rows(x, "hiera_attn4 x")
    if x.dtype != torch.float32 or h.dtype != torch.float16 or tuple(h.shape) != (rows, D) or not h.is_contiguous() or rows != n_img * Gh * Gw:
        raise LmxError("hiera_attn4: h must be contiguous float16 [n*Gh*Gw, D] and x float32 rows of the same count")
    if tuple(img.shape) != (4 * heads, 16384) or img.dtype != torch.float16 or not img.is_contiguous() or bias.numel() != heads * 192 + D \
            or bias.dtype != torch.float32:
        raise LmxError("hiera_attn4: packed operands have the wrong shapes (lmx.sam.pack_hiera_attn4)")
    check(_lib.load().lmx_k_hiera_attn4(_ptr(h), _ptr(x), ldx, _ptr(img), _ptr(bias), n_img, Gh, Gw, D, heads, float((D // heads) ** -0.5),
                                        _stream(dev)), "lmx_k_hiera_attn4")
    return x


def hiera_attn_pool_ok(Din, Dout, heads, win, Gh, Gw, q_stride):
    """Shapes lmx_k_hiera_attn_pool is built for: the blocks that open Hiera-B+ stage 2 (112 -> 224, 4 heads, 8 x 8 windows) and
    stage 3 (224 -> 448, 8 heads, 4 x 4 windows), pooled queries and shortcut."""
    if not q_stride or os.environ.get("LMX_HIERA_ATTN_POOL", "1") == "0":
        return False
    if (Din, Dout, heads, win) == (112, 224, 4, 8):
        return Gh % 8 == 0 and Gw % 8 == 0
    if (Din, Dout, heads, win) == (224, 448, 8, 4):  # the block that opens stage 3 (LMX_HIERA_ATTN_POOL3=0: its separate launches)
        return Gh % 4 == 0 and Gw % 4 == 0 and os.environ.get("LMX_HIERA_ATTN_POOL3", "1") != "0"
    return False


def hiera_attn_pool(h, packed, n_img, Gh, Gw, heads, Dout):
    """The attention half of Hiera's stage-opening block as one launch (csrc/hiera.hip): returns f32 [n*(Gh/2)*(Gw/2), Dout] =
    pool(proj(h)) + attn_proj(window attention(pool(q(h)), k(h), v(h))).  h f16 [n*Gh*Gw, Din] contiguous = layer_norm1(x);
    packed = (w_img, bias) from lmx.sam.pack_hiera_attn_pool."""
    img, bias = packed
    dev = _dev(h, img, bias)
    if h.dtype != torch.float16 or h.dim() != 2 or not h.is_contiguous() or h.shape[0] != n_img * Gh * Gw:
        raise LmxError("hiera_attn_pool: h must be contiguous float16 [n*Gh*Gw, Din]")
    Din = h.shape[1]
    nimg, nb = (47, Dout + heads * 192) if Din > 128 else (14, 2 * Dout + heads * 192)
    if tuple(img.shape) != (nimg, 16384) or img.dtype != torch.float16 or not img.is_contiguous() or bias.numel() != nb or bias.dtype != torch.float32:
        raise LmxError("hiera_attn_pool: packed operands have the wrong shapes (lmx.sam.pack_hiera_attn_pool)")
    out = torch.empty((n_img * (Gh // 2) * (Gw // 2), Dout), dtype=torch.float32, device=h.device)
    check(_lib.load().lmx_k_hiera_attn_pool(_ptr(h), _ptr(out), _ptr(img), _ptr(bias), n_img, Gh, Gw, Din, Dout, heads,
                                            float((Dout // heads) ** -0.5), _stream(dev)), "lmx_k_hiera_attn_pool")
    return out


def ln_mlp_img_ok(D, rows):
    """Where lmx_k_ln_mlp_img (the streamed-image form of the fused LN + MLP, csrc/hiera.hip) replaces csrc/mlp.hip's kernel: D = 112 /
    224 at EVERY row count.  It is faster only from ~1.3 M / 0.33 M rows (tools/mlp_probe.py) and 10 - 30 % slower below a third of
    that, but the two kernels sum in different orders, and a frame's result must not depend on the batch it rides in (DESIGN.md section
    3, Reproducibility: the multi-GPU JSON equals the single-GPU one whatever the sharding) — so one kernel serves all batch sizes.
    LMX_MLP_IMG=0: csrc/mlp.hip's kernel everywhere."""
    return D in (112, 224) and os.environ.get("LMX_MLP_IMG", "1") != "0"


def ln_mlp_img(x, packed, eps, x16=None, h_next=None):
    """x (f32 [rows, D], in place) += fc2(gelu(fc1(LayerNorm(x)))) with the weights as LDS images (lmx.sam.pack_ln_mlp: (w_img, bias);
    bias carries layer_norm2's and — when h_next is given — the next block's layer_norm1's vectors).  x16 / h_next: as ln_mlp."""
    img, bias = packed
    dev = _dev(x, img, bias, x16, h_next)
    rows, D, ldx = _rows(x, "ln_mlp_img x")
    nimg = 7 if D == 112 else 28
    if x.dtype != torch.float32 or tuple(img.shape) != (nimg, 16384) or img.dtype != torch.float16 or not img.is_contiguous() \
            or bias.numel() != 9 * D or bias.dtype != torch.float32:
        raise LmxError("ln_mlp_img: x must be f32 [rows, D] and the packed operands those of lmx.sam.pack_ln_mlp")
    for t, name in ((x16, "x16"), (h_next, "h_next")):
        if t is not None and (t.dtype != torch.float16 or tuple(t.shape) != (rows, D) or not t.is_contiguous()):
            raise LmxError(f"ln_mlp_img: {name} must be contiguous float16 [rows, D]")
    check(_lib.load().lmx_k_ln_mlp_img(_ptr(x), ldx, _ptr(img), _ptr(bias), rows, D, float(eps), _ptr(x16), _ptr(h_next), _stream(dev)),
          "lmx_k_ln_mlp_img")
    return x


def _attn_desc(q, k, v, out, B, H, Tq, Tk, hd, scale, window, pad_k, pad_v):
    d = AttnDesc()
    d.Q, d.K, d.V, d.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    for t in (q, k, v, out):
        if t.dtype != torch.float16 or t.stride(-1) != 1:
            raise LmxError("attention: q/k/v/out must be float16 with contiguous last dim")
    d.B, d.H, d.Tq, d.Tk, d.hd = B, H, Tq, Tk, hd
    d.scale = float(scale)
    if window is None:
        d.mode = 0
    else:
        d.mode = 1
        d.Gh, d.Gw, d.ws, d.q_stride = window["Gh"], window["Gw"], window["ws"], window.get("q_stride", 1)
        d.pad_k = pad_k.data_ptr() if pad_k is not None else None
        d.pad_v = pad_v.data_ptr() if pad_v is not None else None
    return d


def attention(q, k, v, out, B, H, Tq, Tk, hd, scale, window=None, pad_k=None, pad_v=None, rel_pos=None):
    """q/k/v/out: 2-D token-row views [rows, >=H*hd] (row stride = ld).  window = None (flat: row = b*T + t) or
    dict(Gh, Gw, ws, q_stride) for in-place window addressing on the token grid.
    rel_pos = (rel_pos_h, rel_pos_w) f32 [2S-1, hd] adds SAM v1's decomposed relative-position bias (S*S == Tk)."""
    dev = _dev(q, k, v, out, pad_k, pad_v)
    d = _attn_desc(q, k, v, out, B, H, Tq, Tk, hd, scale, window, pad_k, pad_v)
    keep = None
    if rel_pos is not None:
        rh, rw = rel_pos
        _dev(rh, rw, q)
        S = (rh.shape[0] + 1) // 2
        if S * S != Tk or Tq != Tk or rh.dtype != torch.float32 or tuple(rh.shape) != (2 * S - 1, hd) or rh.shape != rw.shape:
            raise LmxError("attention: rel_pos tables must be float32 [2S-1, hd] with S*S == Tq == Tk")
        keep = torch.empty((B * H * Tq, 2 * S), dtype=torch.float16, device=q.device)
        check(_lib.load().lmx_k_relpos_tables(C.byref(d), _ptr(rh), _ptr(rw), S, _ptr(keep), _stream(dev)), "lmx_k_relpos_tables")
        d.rel, d.rel_S = keep.data_ptr(), S
    check(_lib.load().lmx_k_attention(C.byref(d), _stream(dev)), "lmx_k_attention")
    return out


def rope(x, B, T, H, hd, n_prefix, cos_t, sin_t):
    dev = _dev(x, cos_t, sin_t)
    check(_lib.load().lmx_k_rope(_ptr(x), x.stride(0), B, T, H, hd, n_prefix, _ptr(cos_t), _ptr(sin_t), _stream(dev)),
          "lmx_k_rope")
    return x


def pil_resize(src, dw, dh, tab_h, tab_v, swap_rb=False):
    """Pillow-exact two-pass resize of u8 [n,sh,sw,3] -> [n,dh,dw,3].  tab_* = (bounds i32 [2*out], kk i32
    [out*ksize], ksize) device tensors from lmx.resample.coeff_tables; None skips that pass (size unchanged)."""
    dev = _dev(src)
    n, sh, sw, c = src.shape
    if c != 3 or src.dtype != torch.uint8 or not src.is_contiguous():
        raise LmxError("pil_resize: src must be contiguous uint8 [n,h,w,3]")
    lib = _lib.load()
    cur = src
    if tab_h is not None:
        bounds, kk, ksize = tab_h
        tmp = torch.empty((n, sh, dw, 3), dtype=torch.uint8, device=src.device)
        check(lib.lmx_k_pil_resize_h(_ptr(cur), _ptr(tmp), n, sh, sw, dw, _ptr(bounds), _ptr(kk), ksize,
                                     1 if swap_rb else 0, _stream(dev)), "lmx_k_pil_resize_h")
        cur = tmp
    elif swap_rb:
        raise LmxError("pil_resize: swap_rb needs the horizontal pass")
    if tab_v is not None:
        bounds, kk, ksize = tab_v
        w = cur.shape[2]
        dst = torch.empty((n, dh, w, 3), dtype=torch.uint8, device=src.device)
        check(lib.lmx_k_pil_resize_v(_ptr(cur), _ptr(dst), n, cur.shape[1], dh, w, _ptr(bounds), _ptr(kk), ksize,
                                     _stream(dev)), "lmx_k_pil_resize_v")
        cur = dst
    return cur


def patchify_norm(img, top, left, gh, gw, P, lut, k_pad=None):
    """-> f16 [n*gh*gw, k_pad or P*P*3] patch matrix (columns beyond P*P*3 zero)."""
    dev = _dev(img, lut)
    n, ih, iw, c = img.shape
    if c != 3 or img.dtype != torch.uint8 or not img.is_contiguous():
        raise LmxError("patchify_norm: img must be contiguous uint8 [n,h,w,3]")
    K = P * P * 3
    ldo = k_pad or K
    alloc = torch.zeros if ldo != K else torch.empty
    out = alloc((n * gh * gw, ldo), dtype=torch.float16, device=img.device)
    check(_lib.load().lmx_k_patchify_norm(_ptr(img), _ptr(out), n, ih, iw, top, left, gh, gw, P, ldo, _ptr(lut),
                                          _stream(dev)), "lmx_k_patchify_norm")
    return out


def assemble_tokens(patch, prefix, pos, B, np_, n_prefix, D):
    dev = _dev(patch, prefix, pos)
    out = torch.empty((B * (np_ + n_prefix), D), dtype=torch.float32, device=patch.device)
    check(_lib.load().lmx_k_assemble_tokens(_ptr(patch), _ptr(prefix), _ptr(pos), _ptr(out), B, np_, n_prefix, D,
                                            _stream(dev)), "lmx_k_assemble_tokens")
    return out


def token_mean(x, B, T, D):
    dev = _dev(x)
    out = torch.empty((B, D), dtype=torch.float32, device=x.device)
    check(_lib.load().lmx_k_token_mean(_ptr(x), _DT[x.dtype], _ptr(out), B, T, D, _stream(dev)), "lmx_k_token_mean")
    return out


def nms(pred, conf, iou=0.7, max_det=300, max_wh=7680.0):
    """pred f32 [n,A,4+nc] -> (boxes [n,max_det,4], scores, cls, src, counts) on device (lmx_k_nms)."""
    dev = _dev(pred)
    if pred.dtype != torch.float32 or pred.dim() != 3 or not pred.is_contiguous():
        raise LmxError("nms: pred must be contiguous float32 [n,A,4+nc]")
    n, A, row = pred.shape
    lib = _lib.load()
    ws = torch.empty((int(lib.lmx_nms_workspace_bytes(n, A)),), dtype=torch.uint8, device=pred.device)
    boxes = torch.zeros((n, max_det, 4), dtype=torch.float32, device=pred.device)
    scores = torch.zeros((n, max_det), dtype=torch.float32, device=pred.device)
    cls = torch.zeros((n, max_det), dtype=torch.int32, device=pred.device)
    src = torch.full((n, max_det), -1, dtype=torch.int32, device=pred.device)
    counts = torch.zeros((n,), dtype=torch.int32, device=pred.device)
    check(lib.lmx_k_nms(_ptr(pred), n, A, row - 4, float(conf), float(iou), max_det, float(max_wh), _ptr(boxes),
                        _ptr(scores), _ptr(cls), _ptr(src), _ptr(counts), _ptr(ws), _stream(dev)), "lmx_k_nms")
    return boxes, scores, cls, src, counts


# ---- YOLO pieces ------------------------------------------------------------------------------------------
def letterbox(frames, geo, tables, swap_rb=True):
    """u8 [n,sh,sw,3] -> u8 [n,oh,ow,3]; geo from lmx.letterbox.geometry, tables = device (xofs, ialpha, yofs, ibeta)."""
    dev = _dev(frames)
    n, sh, sw, c = frames.shape
    if c != 3 or frames.dtype != torch.uint8 or not frames.is_contiguous():
        raise LmxError("letterbox: frames must be contiguous uint8 [n,h,w,3]")
    out = torch.empty((n, geo.oh, geo.ow, 3), dtype=torch.uint8, device=frames.device)
    xo, ia, yo, ib = tables if tables is not None else (None, None, None, None)
    check(_lib.load().lmx_k_letterbox(_ptr(frames), _ptr(out), n, sh, sw, geo.rh, geo.rw, geo.top, geo.left, geo.oh, geo.ow,
                                      _ptr(xo), _ptr(ia), _ptr(yo), _ptr(ib), 1 if swap_rb else 0, _stream(dev)),
          "lmx_k_letterbox")
    return out


def stem_conv(img_u8, w, bias, out=None):
    dev = _dev(img_u8, w, bias, out)
    n, H, W, _ = img_u8.shape
    Cout = bias.numel()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((n, Ho, Wo, Cout), dtype=torch.float16, device=img_u8.device)
    if not out.is_contiguous() or tuple(out.shape) != (n, Ho, Wo, Cout):
        raise LmxError("stem_conv: out must be a dense NHWC tensor")
    check(_lib.load().lmx_k_stem_conv(_ptr(img_u8), _ptr(w), _ptr(bias), _ptr(out), n, H, W, Cout, _stream(dev)),
          "lmx_k_stem_conv")
    return out


def maxpool5(x, out):
    dev = _dev(x, out)
    n, H, W, Cc, ps = _nhwc(x, "maxpool5 x")
    no, Ho, Wo, Co, pso = _nhwc(out, "maxpool5 out")
    if (no, Ho, Wo, Co) != (n, H, W, Cc):
        raise LmxError("maxpool5: shape mismatch")
    check(_lib.load().lmx_k_maxpool5(_ptr(x), ps, _ptr(out), pso, n, H, W, Cc, _stream(dev)), "lmx_k_maxpool5")
    return out


def upsample2(x, out):
    dev = _dev(x, out)
    n, H, W, Cc, ps = _nhwc(x, "upsample2 x")
    no, Ho, Wo, Co, pso = _nhwc(out, "upsample2 out")
    if (no, Ho, Wo, Co) != (n, 2 * H, 2 * W, Cc):
        raise LmxError("upsample2: shape mismatch")
    check(_lib.load().lmx_k_upsample2(_ptr(x), ps, _ptr(out), pso, n, H, W, Cc, _stream(dev)), "lmx_k_upsample2")
    return out


# ---- the exact plan's x3 format (csrc/exact.hip): a value travels as the f16 channel triple [hi | lo | hi] per group ----------
def split3(x, act, out3, g=None, res3=None):
    """f32 NHWC x [n,H,W,N] (channel slices allowed) -> x3 groups of width g written into out3 [n,H,W,3N] (a channel slice of
    an x3 buffer); y = act(x) + value(res3) (lmx_k_split3).  x may be the [S, n,H,W,N] partial outputs of a split-K launch:
    they are added up in index order first."""
    dev = _dev(x, out3, res3)
    nsum, sstride = 1, 0
    if x.dim() == 5:
        nsum, sstride = x.shape[0], x.stride(0)
        x = x[0]
    n, H, W, N, ps = _nhwc(x, "split3 x")
    no, Ho, Wo, C3, pso = _nhwc(out3, "split3 out")
    g = g or N
    if x.dtype != torch.float32 or out3.dtype != torch.float16 or (no, Ho, Wo, C3) != (n, H, W, 3 * N):
        raise LmxError(f"split3: x f32 {tuple(x.shape)} -> out3 f16 {tuple(out3.shape)} (3x the channels) expected")
    ldr = 0
    if res3 is not None:
        nr, Hr, Wr, Cr, ldr = _nhwc(res3, "split3 res")
        if (nr, Hr, Wr, Cr) != (n, H, W, 3 * N) or res3.dtype != torch.float16:
            raise LmxError("split3: residual must be an x3 tensor of the output's shape")
    check(_lib.load().lmx_k_split3(_ptr(x), ps, act, _ptr(res3), ldr, _ptr(out3), pso, n * H * W, N, g, nsum, sstride, _stream(dev)),
          "lmx_k_split3")
    return out3


def maxpool5_x3(x3, out3):
    dev = _dev(x3, out3)
    n, H, W, C3, ps = _nhwc(x3, "maxpool5_x3 x")
    no, Ho, Wo, Co, pso = _nhwc(out3, "maxpool5_x3 out")
    if (no, Ho, Wo, Co) != (n, H, W, C3) or C3 % 3:
        raise LmxError("maxpool5_x3: shape mismatch")
    check(_lib.load().lmx_k_maxpool5_x3(_ptr(x3), ps, _ptr(out3), pso, n, H, W, C3 // 3, _stream(dev)), "lmx_k_maxpool5_x3")
    return out3


def stem_conv_x3(img_u8, w, bias):
    dev = _dev(img_u8, w, bias)
    n, H, W, _ = img_u8.shape
    Cout = bias.numel()
    out = torch.empty((n, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 3 * Cout), dtype=torch.float16, device=img_u8.device)
    check(_lib.load().lmx_k_stem_conv_x3(_ptr(img_u8), _ptr(w), _ptr(bias), _ptr(out), n, H, W, Cout, _stream(dev)),
          "lmx_k_stem_conv_x3")
    return out


def detect_decode(head, pred, nc, stride, a_off):
    """head f32 [n,H,W,ldh] dense, pred f32 [n,A,4+nc] dense."""
    dev = _dev(head, pred)
    n, H, W, ldh = head.shape
    A = pred.shape[1]
    if not head.is_contiguous() or not pred.is_contiguous() or pred.shape[2] != 4 + nc:
        raise LmxError("detect_decode: head/pred must be dense")
    check(_lib.load().lmx_k_detect_decode(_ptr(head), ldh, _ptr(pred), n, H, W, nc, float(stride), a_off, A, _stream(dev)),
          "lmx_k_detect_decode")
    return pred


def scale_boxes(boxes, padx, pady, gain, w, h):
    dev = _dev(boxes)
    if boxes.dtype != torch.float32 or not boxes.is_contiguous() or boxes.shape[-1] != 4:
        raise LmxError("scale_boxes: boxes must be dense float32 [...,4]")
    check(_lib.load().lmx_k_scale_boxes(_ptr(boxes), boxes.numel() // 4, float(padx), float(pady), float(gain), float(w),
                                        float(h), _stream(dev)), "lmx_k_scale_boxes")
    return boxes


def pose_gather(raws, strides, src, counts, kpt_shape, padx, pady, gain, w, h):
    """Keypoints of the detections NMS kept: raws = the three levels' cv4 outputs f32 [n,h,w,ldk]; src / counts from nms().
    -> f32 [n, max_det, K, ndim] in frame pixels (lmx_k_pose_gather)."""
    dev = _dev(*raws, src, counts)
    K_, ndim = kpt_shape
    ldk = raws[0].shape[-1]
    for r in raws:
        if r.dtype != torch.float32 or not r.is_contiguous() or r.dim() != 4 or r.shape[-1] != ldk:
            raise LmxError("pose_gather: levels must be dense float32 [n,h,w,ldk]")
    if src.dtype != torch.int32 or counts.dtype != torch.int32 or not src.is_contiguous():
        raise LmxError("pose_gather: src / counts must be int32")
    n, max_det = src.shape
    hw = (C.c_int32 * 6)(*[v for r in raws for v in (r.shape[1], r.shape[2])])
    st = (C.c_float * 3)(*[float(v) for v in strides])
    out = torch.empty((n, max_det, K_, ndim), dtype=torch.float32, device=src.device)
    check(_lib.load().lmx_k_pose_gather(_ptr(raws[0]), _ptr(raws[1]), _ptr(raws[2]), ldk, C.cast(hw, C.c_void_p),
                                        C.cast(st, C.c_void_p), _ptr(src), _ptr(counts), n, max_det, K_, ndim, float(padx),
                                        float(pady), float(gain), float(w), float(h), _ptr(out), _stream(dev)), "lmx_k_pose_gather")
    return out


def add_bcast(a, b, out=None, out_dtype=torch.float32):
    """out[r] = a[r] + b[r % b_rows]  (a, b float32 2-D; out float32 or float16)."""
    dev = _dev(a, b, out)
    rows, D_, lda = _rows(a, "add_bcast a")
    b_rows, Db, ldb = _rows(b, "add_bcast b")
    if Db != D_ or a.dtype not in _DT or b.dtype != torch.float32:
        raise LmxError("add_bcast: a must be float16/float32, b float32, equal width")
    if out is None:
        out = torch.empty((rows, D_), dtype=out_dtype, device=a.device)
    check(_lib.load().lmx_k_add_bcast(_ptr(a), _DT[a.dtype], lda, _ptr(b), ldb, b_rows, _ptr(out), _DT[out.dtype], out.stride(0),
                                      rows, D_, _stream(dev)), "lmx_k_add_bcast")
    return out


def attention_f32(q, k, v, B, H, Tq, Tk, hd, scale):
    """softmax(q k^T * scale) v in plain f32 for the SAM decoder's tiny attentions (lmx_k_attention_f32): q [B*Tq, >=H*hd],
    k / v [B*Tk, >=H*hd] f32 row views -> f32 [B*Tq, H*hd]."""
    dev = _dev(q, k, v)
    for t in (q, k, v):
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise LmxError("attention_f32: q/k/v must be float32 2-D row views")
    out = torch.empty((B * Tq, H * hd), dtype=torch.float32, device=q.device)
    check(_lib.load().lmx_k_attention_f32(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(out), out.stride(0),
                                          B, H, Tq, Tk, hd, float(scale), _stream(dev)), "lmx_k_attention_f32")
    return out


def hyper_mask_f32(up, hyper, n, G, C_, act=ACT_NONE):
    dev = _dev(up, hyper)
    if up.dtype != torch.float32 or not up.is_contiguous():
        raise LmxError("hyper_mask_f32: up must be contiguous float32")
    logits = torch.empty((n, 4 * G, 4 * G), dtype=torch.float32, device=up.device)
    check(_lib.load().lmx_k_hyper_mask_f32(_ptr(up), _ptr(hyper), _ptr(logits), n, G, C_, act, _stream(dev)), "lmx_k_hyper_mask_f32")
    return logits


def split3_rows(x, act=ACT_NONE):
    """f32 [rows, N] (row stride allowed) -> contiguous x3 rows f16 [rows, 3N] = [hi | lo | hi] (one group): the A operand of an
    exact Linear (lmx_k_split3)."""
    dev = _dev(x)
    rows, N, ldx = _rows(x, "split3_rows x")
    if x.dtype != torch.float32 or N % 8:
        raise LmxError("split3_rows: float32 rows with N % 8 == 0 expected")
    out = torch.empty((rows, 3 * N), dtype=torch.float16, device=x.device)
    check(_lib.load().lmx_k_split3(_ptr(x), ldx, act, None, 0, _ptr(out), 3 * N, rows, N, N, 1, 0, _stream(dev)), "lmx_k_split3")
    return out


def prompt_box(boxes, sx, sy, S, gauss, corner):
    """boxes f32 [n,>=4] (row stride allowed) in frame pixels -> sparse f32 [n,2,2F] (lmx_k_prompt_box)."""
    dev = _dev(boxes, gauss, corner)
    n = boxes.shape[0]
    Fq = gauss.shape[1]
    out = torch.empty((n, 2, 2 * Fq), dtype=torch.float32, device=boxes.device)
    check(_lib.load().lmx_k_prompt_box(_ptr(boxes), boxes.stride(0), _ptr(out), n, float(sx), float(sy), float(S), _ptr(gauss),
                                       _ptr(corner), Fq, _stream(dev)), "lmx_k_prompt_box")
    return out


def hyper_mask(up, hyper, n, G, C_):
    dev = _dev(up, hyper)
    logits = torch.empty((n, 4 * G, 4 * G), dtype=torch.float32, device=up.device)
    check(_lib.load().lmx_k_hyper_mask(_ptr(up), _ptr(hyper), _ptr(logits), n, G, C_, _stream(dev)), "lmx_k_hyper_mask")
    return logits


def mask_post(logits, T, nh, nw, h, w):
    """-> (mask u8 [n,h,w], stats int64 [n,8] = area, sum_x, sum_y, min_x, min_y, max_x, max_y, 0)."""
    dev = _dev(logits)
    n, L, _ = logits.shape
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=logits.device)
    stats = torch.empty((n, 8), dtype=torch.int64, device=logits.device)
    ws = torch.empty((n, nh, nw), dtype=torch.float32, device=logits.device)
    check(_lib.load().lmx_k_mask_post(_ptr(logits), n, L, T, nh, nw, h, w, _ptr(mask), _ptr(stats), _ptr(ws), _stream(dev)),
          "lmx_k_mask_post")
    return mask, stats


def contour_features(mask):
    """mask u8 [n,h,w] (0 / non-0) on device -> int64 [n,8] = 2*contourArea, unit steps, diagonal steps, min x, min y, max x,
    max y, number of external contours of the largest external contour (lmx_k_contour_features, csrc/contour.hip)."""
    dev = _dev(mask)
    if mask.dtype != torch.uint8 or mask.dim() != 3 or not mask.is_contiguous():
        raise LmxError("contour_features: mask must be contiguous uint8 [n,h,w]")
    n, h, w = mask.shape
    lib = _lib.load()
    ws = torch.empty((int(lib.lmx_contour_workspace_bytes(n, h, w)),), dtype=torch.uint8, device=mask.device)
    out = torch.empty((n, 8), dtype=torch.int64, device=mask.device)
    check(lib.lmx_k_contour_features(_ptr(mask), n, h, w, _ptr(out), _ptr(ws), _stream(dev)), "lmx_k_contour_features")
    return out


# ---- optional per-launch timing (bench.py's roofline leg) -----------------------------------------------------------
# While a trace is active every lmx_k_* entry point is bracketed by a HIP event pair recorded on the stream the launch goes
# to (torch's current stream of the operands' device), together with the ALGORITHMIC work of the call: flops and the minimal
# bytes its operands and results occupy (no re-reads, no workspace) - DESIGN.md section 3 states the per-unit figures.
LAUNCH_TRACE = None
_raw_lib = None
RIDGE_FLOP_PER_BYTE = 2500e12 / 8e12  # dense f16 MFMA peak / HBM peak (MI355X_MICROARCH.md): above it a launch is MFMA-bound


def _work(name, args):
    """(class, flops, bytes) of one C-ABI call; bytes None = not modelled (pre/post-processing glue: time share only)."""
    if name == "lmx_k_gemm":
        d = args[0]._obj
        M, N, Kd = d.M, d.N, d.K
        osz = 4 if d.out_dtype == F32 else 2
        a_bytes = 2 * M * (Kd // max(d.a_rep, 1))
        if d.a_mode == 1:  # 3x3 implicit GEMM: the input image is read once, not 9 times
            a_bytes = 2 * (M // max(d.Ho * d.Wo, 1)) * d.H * d.W_ * d.Cin
        by = a_bytes + 2 * N * Kd + osz * (M // 4 if d.a_mode == 2 else M) * N + (osz * (d.res_rows or M) * N if d.res else 0)
        fl = 2.0 * M * N * Kd
        key = f"gemm M={M} N={N} K={Kd} out={'f32' if osz == 4 else 'f16'} conv3x3={d.a_mode} act={d.act} res={int(bool(d.res))}"
        return ("gemm/mfma-bound" if fl / by >= RIDGE_FLOP_PER_BYTE else "gemm/hbm-bound"), fl, by, key
    if name == "lmx_k_attention":
        d = args[0]._obj
        fl = 4.0 * d.B * d.H * d.Tq * d.Tk * d.hd
        by = 2 * d.H * d.hd * d.B * (2 * d.Tq + 2 * d.Tk)
        key = f"attention B={d.B} H={d.H} Tq={d.Tq} Tk={d.Tk} hd={d.hd} mode={d.mode} ws={d.ws} qs={d.q_stride} rel={int(bool(d.rel))}"
        return ("attention/mfma-bound" if fl / by >= RIDGE_FLOP_PER_BYTE else "attention/hbm-bound"), fl, by, key
    if name == "lmx_k_layernorm":
        in_dt, out_dt, rows, D = args[1], args[6], args[8], args[9]
        return "layernorm", 0.0, rows * D * ((4 if in_dt == F32 else 2) + (4 if out_dt == F32 else 2)), f"layernorm rows={rows} D={D} in={in_dt} out={out_dt}"
    if name == "lmx_k_ln_mlp_img":
        rows, D = args[4], args[5]
        return "fused ln+mlp", 16.0 * D * D * rows, (8 + (2 if args[7] else 0) + (2 if args[8] else 0)) * D * rows, f"ln_mlp_img rows={rows} D={D}"
    if name == "lmx_k_ln_mlp":
        rows, D = args[8], args[9]
        return "fused ln+mlp", 16.0 * D * D * rows, (16 + (2 if args[12] else 0) + (2 if args[15] else 0)) * D * rows, f"ln_mlp rows={rows} D={D}"
    # the attention half of a Hiera block as one launch (csrc/hiera.hip): algorithmic flops of the unpadded products, bytes = the
    # f32 stream read + written (+ the f16 LayerNorm rows where they are an input)
    if name == "lmx_k_hiera_attn8":
        n_, Gh, Gw, D = args[10], args[11], args[12], args[13]
        rows = n_ * Gh * Gw
        return "fused attention half", rows * (8.0 * D * D + 256.0 * D), rows * D * (8 + (2 if args[0] else 0)), f"hiera_attn8 rows={rows} D={D} ln_inside={int(not args[0])}"
    if name == "lmx_k_hiera_attn4":
        n_, Gh, Gw, D = args[5], args[6], args[7], args[8]
        rows = n_ * Gh * Gw
        return "fused attention half", rows * (8.0 * D * D + 64.0 * D), rows * D * 10, f"hiera_attn4 rows={rows} D={D}"
    if name == "lmx_k_hiera_attn_pool":
        n_, Gh, Gw, Di, Do = args[4], args[5], args[6], args[7], args[8]
        rows = n_ * Gh * Gw
        keys = 64 if Di == 112 else 16  # tokens of a window
        return "fused attention half", rows * (8.0 * Di * Do + keys * Do + 0.5 * Do * Do), rows * (2 * Di + Do), f"hiera_attn_pool rows={rows} {Di}->{Do}"
    # streaming element-wise glue with a plain byte count: one HBM-bound class of its own (the rest — resizes, im2col, NMS,
    # mask_post, contour features, decode — stays "pre/post-processing and glue": time share only)
    if name == "lmx_k_cast_f32_f16":
        rows, cols = args[4], args[5]
        return "streaming glue", 0.0, 6 * rows * cols, f"cast_f32_f16 rows={rows} cols={cols}"
    if name == "lmx_k_maxpool2":
        dt, n_, H, W, C_ = args[4], args[5], args[6], args[7], args[8]
        esz = 4 if dt == F32 else 2
        return "streaming glue", 0.0, esz * n_ * H * W * C_ * 5 // 4, f"maxpool2 n={n_} H={H} W={W} C={C_} dtype={dt}"
    if name == "lmx_k_upsample2":
        n_, H, W, C_ = args[4], args[5], args[6], args[7]
        return "streaming glue", 0.0, 2 * n_ * H * W * C_ * 5, f"upsample2 n={n_} H={H} W={W} C={C_}"
    if name == "lmx_k_add_bcast":
        a_dt, o_dt, rows, D = args[1], args[7], args[9], args[10]
        return "streaming glue", 0.0, rows * D * ((4 if a_dt == F32 else 2) + (4 if o_dt == F32 else 2)), f"add_bcast rows={rows} D={D}"
    if name == "lmx_k_rope":
        B, T, H, hd = args[2], args[3], args[4], args[5]
        return "streaming glue", 0.0, 2 * 2 * 2 * B * T * H * hd, f"rope B={B} T={T} H={H} hd={hd}"  # q and k, read + written, f16
    return "pre/post-processing and glue", 0.0, None, name


def _install_trace():
    global _raw_lib
    if _raw_lib is not None:
        return
    lib = _raw_lib = _lib.load()

    def make(name, fn):
        def traced(*args):
            if LAUNCH_TRACE is None:
                return fn(*args)
            cls, fl, by, key = _work(name, args)
            st = torch.cuda.current_stream(_tls.dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = fn(*args)
            e1.record(st)
            LAUNCH_TRACE.append((cls, key, fl, by, e0, e1))
            return rc

        return traced

    class _Proxy:
        def __getattr__(self, name):
            fn = getattr(lib, name)
            if name.startswith("lmx_k_"):
                fn = make(name, fn)
            setattr(self, name, fn)
            return fn

    _lib._lib = _Proxy()


def start_launch_trace():
    global LAUNCH_TRACE
    _install_trace()
    LAUNCH_TRACE = []


def stop_launch_trace(by_shape=False):
    """-> {class: dict(launches, seconds, flops, bytes)} over the traced launches (events read after a device sync);
    by_shape=True keys the table by (class, operand shape) instead."""
    global LAUNCH_TRACE
    tr, LAUNCH_TRACE = LAUNCH_TRACE or [], None
    torch.cuda.synchronize()
    out, shapes = {}, {}
    for cls, key, fl, by, e0, e1 in tr:
        dt = e0.elapsed_time(e1) * 1e-3
        for table, k in ((out, cls), (shapes, (cls, key))):
            r = table.setdefault(k, dict(launches=0, seconds=0.0, flops=0.0, bytes=0.0, modelled=by is not None))
            r["launches"] += 1
            r["seconds"] += dt
            r["flops"] += fl
            r["bytes"] += by or 0.0
    return (out, shapes) if by_shape else out
