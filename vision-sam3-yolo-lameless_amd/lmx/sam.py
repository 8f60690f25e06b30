"""SAM image encoder on liblmx — the ``predictor.set_image(image)`` half of services/sam3-pipeline/app/main.py:80.

BASELINE cfg#3 names the SAM2 Hiera-B+ trunk + FPN neck; this module runs it as a launch
sequence over the C-ABI kernels:
  pre   : PIL-bilinear ResizeLongestSide(1024) on u8 (lmx_k_pil_resize_*), SamPredictor's (x-mean)/std applied
          to the frame AS GIVEN (the service hands over BGR: SURVEY Appendix C-2) and the zero pad to 1024^2 are
          folded into the patch-embed im2col (lmx_k_im2col_u8);
  trunk : patch-embed GEMM with the windowed position table added in the epilogue (residual broadcast), then 24
          multi-scale blocks: LN -> qkv GEMM -> [2x2 max Q-pool] -> window / global flash attention addressed IN PLACE
          on the token grid (no window_partition copies; padded keys take the qkv bias) -> proj GEMM (+residual, or
          + pooled `proj` shortcut at stage changes) -> LN -> MLP GEMMs (GELU, +residual).  f32 residual stream.
  neck  : 1x1 lateral GEMMs to 256 channels, nearest-x2 top-down add on levels 2/3 (GEMM residual epilogue).
"""
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import kernels as K
from . import resample

SAM_PIXEL_MEAN = (123.675, 116.28, 103.53)
SAM_PIXEL_STD = (58.395, 57.12, 57.375)


@dataclass
class HieraConfig:
    hidden: int = 112
    blocks: tuple = (2, 3, 16, 3)
    dims: tuple = (112, 224, 448, 896)
    heads: tuple = (2, 4, 8, 16)
    windows: tuple = (8, 4, 14, 7)
    global_blocks: tuple = (12, 16, 20)
    pos_bkg: tuple = (14, 14)
    q_pool_stages: int = 3
    fpn_dim: int = 256
    fpn_top_down: tuple = (2, 3)
    eps: float = 1e-6
    image: int = 1024

    def block_plan(self):
        """[(dim_in, dim_out, heads, window, q_stride)] per block (TF sam2 :466-486)."""
        plan, t = [], 0
        for s, nb in enumerate(self.blocks):
            for b in range(nb):
                first = s > 0 and b == 0
                dim = self.dims[s - 1] if first else self.dims[s]
                win = self.windows[s - 1] if first else self.windows[s]
                if t in self.global_blocks:
                    win = 0
                qs = 2 if (0 < s <= self.q_pool_stages and b == 0) else 0
                plan.append((dim, self.dims[s], self.heads[s], win, qs))
                t += 1
        return plan


def hiera_b_plus():
    return HieraConfig()


def param_spec(cfg):
    """Ordered {transformers Sam2VisionModel parameter name: (shape, init kind)}."""
    s = {}
    s["backbone.pos_embed"] = ((1, cfg.hidden) + tuple(cfg.pos_bkg), "tok")
    s["backbone.pos_embed_window"] = ((1, cfg.hidden, cfg.windows[0], cfg.windows[0]), "tok")
    s["backbone.patch_embed.projection.weight"] = ((cfg.hidden, 3, 7, 7), "w")
    s["backbone.patch_embed.projection.bias"] = ((cfg.hidden,), "b")
    for i, (dim, dim_out, heads, win, qs) in enumerate(cfg.block_plan()):
        p = f"backbone.blocks.{i}."
        s[p + "layer_norm1.weight"] = ((dim,), "g")
        s[p + "layer_norm1.bias"] = ((dim,), "b")
        s[p + "attn.qkv.weight"] = ((3 * dim_out, dim), "w")
        s[p + "attn.qkv.bias"] = ((3 * dim_out,), "b")
        s[p + "attn.proj.weight"] = ((dim_out, dim_out), "w")
        s[p + "attn.proj.bias"] = ((dim_out,), "b")
        s[p + "layer_norm2.weight"] = ((dim_out,), "g")
        s[p + "layer_norm2.bias"] = ((dim_out,), "b")
        s[p + "mlp.proj_in.weight"] = ((4 * dim_out, dim_out), "w")
        s[p + "mlp.proj_in.bias"] = ((4 * dim_out,), "b")
        s[p + "mlp.proj_out.weight"] = ((dim_out, 4 * dim_out), "w")
        s[p + "mlp.proj_out.bias"] = ((dim_out,), "b")
        if dim != dim_out:
            s[p + "proj.weight"] = ((dim_out, dim), "w")
            s[p + "proj.bias"] = ((dim_out,), "b")
    for j, c in enumerate(reversed(cfg.dims)):
        s[f"neck.convs.{j}.weight"] = ((cfg.fpn_dim, c, 1, 1), "w")
        s[f"neck.convs.{j}.bias"] = ((cfg.fpn_dim,), "b")
    return s


def resize_longest_side(h, w, target=1024):
    """segment_anything ResizeLongestSide.get_preprocess_shape: int(x*scale + 0.5)."""
    scale = target * 1.0 / max(h, w)
    return int(h * scale + 0.5), int(w * scale + 0.5)


def sam_norm_lut():
    """lut[c][u] = (f32(u) - mean[c]) / std[c] — Sam.preprocess' `(x - pixel_mean) / pixel_std` on the f32 image."""
    u = np.arange(256, dtype=np.float32)
    m = np.array(SAM_PIXEL_MEAN, np.float32)
    s = np.array(SAM_PIXEL_STD, np.float32)
    return ((u[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def pack_hiera_attn(wqkv, bqkv, wo, bo, heads, ln_inside=False):
    """Operands of lmx_k_hiera_attn8 (csrc/hiera.hip) from the block's torch-layout parameters: wqkv [3D, D], bqkv [3D], wo [D, D],
    bo [D] (numpy, f32).  Returns (wqkv_p f16 [3*heads*64, 128], bqkv_p f32 [3*heads*64], wo_p f16 [D, heads*64], bo f32 [D]):
    q | k | v sections with each head padded from D/heads to 64 rows; v's row 63 of every head is zero with bias 1 (the softmax sum
    then rides the PV product); the 64 columns of a head in wo_p — and, with ln_inside (the kernel normalises the f32 rows itself
    and holds them in accumulator layout), the 128 input columns of wqkv_p — are in MFMA k-slot order: position 32 s + 8 g + 4 h + i
    holds feature 16 (2 s + h) + 4 g + i, the order in which an accumulator tile is an operand."""
    D = wo.shape[0]
    hd = D // heads
    if hd > 63 or D > 128:
        raise ValueError("pack_hiera_attn: head dim <= 63 and D <= 128")
    wq = np.zeros((3 * heads * 64, 128), np.float32)
    bq = np.zeros((3 * heads * 64,), np.float32)
    pos = np.array([32 * s_ + 8 * g + 4 * hb + i for s_ in range(4) for hb in range(2) for g in range(4) for i in range(4)])  # of feature 16(2s+hb)+4g+i
    pos = pos[:D] if ln_inside else np.arange(D)
    for sec in range(3):
        for hh in range(heads):
            r0 = sec * heads * 64 + hh * 64
            wq[r0:r0 + hd, pos] = wqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd]
            bq[r0:r0 + hd] = bqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd]
            if sec == 2:
                bq[r0 + 63] = 1.0
    wop = np.zeros((D, heads * 64), np.float32)
    for hh in range(heads):
        for s_ in range(2):
            for g in range(4):
                for hb in range(2):
                    for i in range(4):
                        d = 16 * (2 * s_ + hb) + 4 * g + i
                        if d < hd:
                            wop[:, 64 * hh + 32 * s_ + 8 * g + 4 * hb + i] = wo[:, hh * hd + d]
    return wq.astype(np.float16), bq, wop.astype(np.float16), np.ascontiguousarray(bo, dtype=np.float32)


def pack_hiera_attn4(wqkv, bqkv, wo, bo, heads):
    """Operands of lmx_k_hiera_attn4 (csrc/hiera.hip): the LDS images of the 4 * heads matrices the kernel streams, and its biases.
    wqkv [3D, D], bqkv [3D], wo [D, D], bo [D] (numpy, f32; D = 224, heads = 4).  Image 4 h + s, s in q | k | v: 64 rows (the
    head's 56, then zeros) of 512 bytes, the 16-byte chunk c of row r stored at chunk c ^ (r & 15); image 4 h + 3: the projection's
    columns of head h as 256 rows (224 outputs, then zeros) of 128 bytes, chunk c of row r at c ^ ((r >> 1) & 7), the 64 columns in
    MFMA k-slot order (position 32 s + 8 g + 4 hb + i holds the head's input 16 (2 s + hb) + 4 g + i, zeros past 56).
    bias: [head][q | k | v][64] (v's entry 63 is 1: the softmax sum rides the PV product) then bo."""
    D = wo.shape[0]
    hd = D // heads
    img = np.zeros((4 * heads, 16384), np.float16)
    bias = np.zeros((heads * 192 + D,), np.float32)
    for hh in range(heads):
        for sec in range(3):
            m = np.zeros((64, 256), np.float16)
            m[:hd, :D] = wqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd].astype(np.float16)
            ch = m.reshape(64, 32, 8)
            out = np.zeros_like(ch)
            for r in range(64):
                out[r, np.arange(32) ^ (r & 15)] = ch[r]
            img[4 * hh + sec] = out.reshape(-1)
            bias[hh * 192 + sec * 64: hh * 192 + sec * 64 + hd] = bqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd]
        bias[hh * 192 + 128 + 63] = 1.0
        m = np.zeros((256, 64), np.float16)
        for s_ in range(2):
            for g in range(4):
                for hb in range(2):
                    for i in range(4):
                        d = 16 * (2 * s_ + hb) + 4 * g + i
                        if d < hd:
                            m[:D, 32 * s_ + 8 * g + 4 * hb + i] = wo[:, hh * hd + d].astype(np.float16)
        ch = m.reshape(256, 8, 8)
        out = np.zeros_like(ch)
        for r in range(256):
            out[r, np.arange(8) ^ ((r >> 1) & 7)] = ch[r]
        img[4 * hh + 3] = out.reshape(-1)
    bias[heads * 192:] = bo
    return img, bias


def _lds_image(m, key):
    """f16 matrix [rows, cols] (cols * 2 bytes = 128 or 256 per row) -> its LDS image: the 16-byte chunk c of row r stored at chunk
    c ^ key(r); zero-padded to 32 KB."""
    rows, cols = m.shape
    ch = m.reshape(rows, cols // 8, 8)
    out = np.zeros_like(ch)
    for r in range(rows):
        out[r, np.arange(cols // 8) ^ key(r)] = ch[r]
    img = np.zeros((16384,), np.float16)
    img[:rows * cols] = out.reshape(-1)
    return img


def pack_hiera_attn_pool(wsc, bsc, wqkv, bqkv, wo, bo, heads):
    """Operands of lmx_k_hiera_attn_pool (csrc/hiera.hip) for a block that opens a stage: wsc [Dout, Din] / bsc: the shortcut's
    projection; wqkv [3 Dout, Din], bqkv; wo [Dout, Dout], bo (numpy, f32).  LDS images of 32 KB, in the order the kernel streams them.
    Din 112 -> Dout 224 (4 heads): 14 images — shortcut rows 0..127, shortcut rows 128.., then per head [q | k] (64 + 64 rows of 256
    bytes, the head's 56 rows then zeros), [v], and the projection's columns of the head (rows of 128 bytes in MFMA k-slot order).
    Din 224 -> Dout 448 (8 heads): 47 images — shortcut in 7 images of 64 rows (512-byte rows), then per head q, k, v (64 rows each) and
    the projection's columns of the head in two images (output rows 0..223, 224..447).  Rows of 256 / 512 bytes are swizzled by r & 15,
    rows of 128 bytes by (r >> 1) & 7.  bias: shortcut + projection [Dout], [head][q | k | v][64] (v's entry 63 is 1)[, Dout zeros]."""
    Dout, Din = wsc.shape
    hd = Dout // heads
    wide = Din > 128  # 512-byte rows
    cols = 256 if wide else 128
    k15, k7 = (lambda r: r & 15), (lambda r: (r >> 1) & 7)

    def rows_img(w, nrows):  # [<= nrows, Din] -> image of nrows rows of `cols` halfs
        m = np.zeros((nrows, cols), np.float16)
        m[:w.shape[0], :Din] = w.astype(np.float16)
        return _lds_image(m, k15)

    if wide:
        imgs = [rows_img(wsc[64 * j: 64 * j + 64], 64) for j in range(7)]
    else:
        imgs = [rows_img(wsc[:128], 128), rows_img(wsc[128:], 128)]
    bias = np.zeros((Dout + heads * 192 + (0 if wide else Dout),), np.float32)
    bias[:Dout] = bsc + bo  # the shortcut's and the output projection's biases: one vector, added once
    for hh in range(heads):
        sec = []
        for s_ in range(3):
            sec.append(wqkv[s_ * Dout + hh * hd: s_ * Dout + (hh + 1) * hd])
            bias[Dout + hh * 192 + s_ * 64: Dout + hh * 192 + s_ * 64 + hd] = bqkv[s_ * Dout + hh * hd: s_ * Dout + (hh + 1) * hd]
        bias[Dout + hh * 192 + 128 + 63] = 1.0
        if wide:
            imgs += [rows_img(sec[0], 64), rows_img(sec[1], 64), rows_img(sec[2], 64)]
        else:
            qk = np.zeros((128, Din), np.float32)
            qk[:hd], qk[64:64 + hd] = sec[0], sec[1]
            imgs += [rows_img(qk, 128), rows_img(sec[2], 64)]
        m = np.zeros((512 if wide else 256, 64), np.float16)
        for s_ in range(2):
            for g in range(4):
                for hb in range(2):
                    for i in range(4):
                        d = 16 * (2 * s_ + hb) + 4 * g + i
                        if d < hd:
                            m[:Dout, 32 * s_ + 8 * g + 4 * hb + i] = wo[:, hh * hd + d].astype(np.float16)
        if wide:
            for half in range(2):
                mm = np.zeros((256, 64), np.float16)
                mm[:224] = m[224 * half: 224 * half + 224]
                imgs.append(_lds_image(mm, k7))
        else:
            imgs.append(_lds_image(m, k7))
    return np.stack(imgs), bias


def pack_ln_mlp(w1, b1, w2, b2, g2, e2, gn=None, en=None):
    """Operands of lmx_k_ln_mlp_img (csrc/hiera.hip): w1 [4D, D], b1 [4D], w2 [D, 4D], b2 [D] (torch Linear layouts), layer_norm2's
    g2 / e2 and — for the kernel's h_next output — the next block's layer_norm1's gn / en (numpy, f32; D = 112 or 224).  Per step of
    64 hidden units: the 64 rows of w1 with their D input columns in MFMA k-slot order (rows of 256 bytes at D = 112, 512 at 224;
    16-byte chunk c of row r at c ^ (r & 15)) and the D rows x 64 columns of w2, columns in k-slot order (rows of 128 bytes, chunk c of
    row r at c ^ ((r >> 1) & 7)); at D = 112 both halves share an image (w2's at byte 16384), at D = 224 they alternate."""
    D = w2.shape[0]
    cols = 128 if D <= 128 else 256
    nks = cols // 32
    kslot = np.array([16 * (2 * s_ + hb) + 4 * g + i for s_ in range(nks) for g in range(4) for hb in range(2) for i in range(4)])  # feature at position
    k15, k7 = (lambda r: r & 15), (lambda r: (r >> 1) & 7)
    imgs = []
    for ch in range(4 * D // 64):
        m1 = np.zeros((64, cols), np.float16)
        ok = kslot < D
        m1[:, ok] = w1[64 * ch: 64 * ch + 64][:, kslot[ok]].astype(np.float16)
        m2 = np.zeros((D, 64), np.float16)
        m2[:, :] = w2[:, 64 * ch + kslot[:64]].astype(np.float16)
        i1, i2 = _lds_image(m1, k15), _lds_image(m2, k7)
        if D <= 128:
            both = i1.copy()
            both[8192:8192 + D * 64] = i2[:D * 64]
            imgs.append(both)
        else:
            imgs += [i1, i2]
    z = np.zeros((D,), np.float32)
    bias = np.concatenate([b1, b2, g2, e2, gn if gn is not None else z, en if en is not None else z]).astype(np.float32)
    return np.stack(imgs), bias


class HieraEncoder:
    """Device-resident Hiera trunk + FPN.  ``encode(frames)`` -> dict(fpn=[3 NHWC f16 levels, high->low res],
    stages=[4 f32 stage outputs]).  Token grids are [n, H, W, C] row-major throughout (no partition copies)."""

    def __init__(self, cfg, state_dict, device="cuda", fused_mlp=True):
        self.cfg = cfg
        self.device = torch.device(device)
        # False: LN / GEMM / GEMM launches for every width (A/B comparisons and tests; LMX_NO_FUSED_MLP=1 forces it)
        self.fused_mlp = fused_mlp and not os.environ.get("LMX_NO_FUSED_MLP")
        dev = self.device
        sd = state_dict

        def t32(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        def t16(a):
            return t32(a).to(torch.float16).contiguous()

        # patch embed: conv [D,3,7,7] -> GEMM [D, (ky,kx,c)] padded to K%8==0
        w = np.transpose(sd["backbone.patch_embed.projection.weight"], (0, 2, 3, 1)).reshape(cfg.hidden, 147)
        self.k_pad = 152
        self.pe_w = t16(np.concatenate([w, np.zeros((cfg.hidden, self.k_pad - 147), np.float32)], 1))
        self.pe_b = t32(sd["backbone.patch_embed.projection.bias"])
        # windowed position table for the (fixed) input grid, computed once on the host like _get_pos_embed (:640-646)
        g = cfg.image // 4
        pe = torch.nn.functional.interpolate(torch.from_numpy(sd["backbone.pos_embed"]), size=(g, g), mode="bicubic")
        win = torch.from_numpy(sd["backbone.pos_embed_window"])
        pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
        self.pos = pe.permute(0, 2, 3, 1).reshape(g * g, cfg.hidden).contiguous().to(dev)
        self.grid0 = g
        self.blocks = []
        for i, (dim, dim_out, heads, win_, qs) in enumerate(cfg.block_plan()):
            p = f"backbone.blocks.{i}."
            qkv_b = sd[p + "attn.qkv.bias"]
            blk = dict(dim=dim, dim_out=dim_out, heads=heads, win=win_, qs=qs,
                       g1=t32(sd[p + "layer_norm1.weight"]), b1=t32(sd[p + "layer_norm1.bias"]),
                       wqkv=t16(sd[p + "attn.qkv.weight"]), bqkv=t32(qkv_b),
                       # what Linear(0) yields for a zero-padded token, rounded like the GEMM output
                       padkv=t16(qkv_b),
                       wo=t16(sd[p + "attn.proj.weight"]), bo=t32(sd[p + "attn.proj.bias"]),
                       g2=t32(sd[p + "layer_norm2.weight"]), b2=t32(sd[p + "layer_norm2.bias"]),
                       w1=t16(sd[p + "mlp.proj_in.weight"]), bb1=t32(sd[p + "mlp.proj_in.bias"]),
                       w2=t16(sd[p + "mlp.proj_out.weight"]), bb2=t32(sd[p + "mlp.proj_out.bias"]))
            if dim != dim_out:
                blk["wp"], blk["bp"] = t16(sd[p + "proj.weight"]), t32(sd[p + "proj.bias"])
            if dim == dim_out == 112 and heads == 2 and win_ == 8 and not qs:  # stage 1 of Hiera-B+: one kernel per attention half
                # (the first block normalises in the kernel; later ones read the rows their predecessor's fused MLP left: see trunk)
                blk["attn8_ln"] = i == 0 or not (self.fused_mlp and dim in K.FUSED_MLP_WIDTHS)
                blk["attn8"] = tuple(torch.from_numpy(a).to(dev) for a in pack_hiera_attn(
                    np.asarray(sd[p + "attn.qkv.weight"], np.float32), np.asarray(qkv_b, np.float32),
                    np.asarray(sd[p + "attn.proj.weight"], np.float32), np.asarray(sd[p + "attn.proj.bias"], np.float32), heads,
                    ln_inside=blk["attn8_ln"]))
            if dim == dim_out == 224 and heads == 4 and win_ == 4 and not qs:  # stage 2 (after its first block): the same, weights streamed
                blk["attn4"] = tuple(torch.from_numpy(a).to(dev) for a in pack_hiera_attn4(
                    np.asarray(sd[p + "attn.qkv.weight"], np.float32), np.asarray(qkv_b, np.float32),
                    np.asarray(sd[p + "attn.proj.weight"], np.float32), np.asarray(sd[p + "attn.proj.bias"], np.float32), heads))
            if qs and ((dim, dim_out, heads, win_) in ((112, 224, 4, 8), (224, 448, 8, 4))):  # the blocks that open stages 2 and 3: pooled queries and shortcut
                blk["attnp"] = tuple(torch.from_numpy(a).to(dev) for a in pack_hiera_attn_pool(
                    np.asarray(sd[p + "proj.weight"], np.float32), np.asarray(sd[p + "proj.bias"], np.float32),
                    np.asarray(sd[p + "attn.qkv.weight"], np.float32), np.asarray(qkv_b, np.float32),
                    np.asarray(sd[p + "attn.proj.weight"], np.float32), np.asarray(sd[p + "attn.proj.bias"], np.float32), heads))
            self.blocks.append(blk)
        for i, blk in enumerate(self.blocks):  # the fused MLP's operands as LDS images, with the NEXT block's layer_norm1 vectors
            if blk["dim_out"] in (112, 224) and self.fused_mlp:
                p = f"backbone.blocks.{i}."
                nx = f"backbone.blocks.{i + 1}." if i + 1 < len(self.blocks) else None
                f32 = lambda k: np.asarray(sd[k], np.float32)  # noqa: E731
                blk["mlp_img"] = tuple(torch.from_numpy(a).to(dev) for a in pack_ln_mlp(
                    f32(p + "mlp.proj_in.weight"), f32(p + "mlp.proj_in.bias"), f32(p + "mlp.proj_out.weight"), f32(p + "mlp.proj_out.bias"),
                    f32(p + "layer_norm2.weight"), f32(p + "layer_norm2.bias"),
                    f32(nx + "layer_norm1.weight") if nx else None, f32(nx + "layer_norm1.bias") if nx else None))
        n = len(cfg.dims) - 1
        self.neck = [(t16(sd[f"neck.convs.{n - i}.weight"][:, :, 0, 0]), t32(sd[f"neck.convs.{n - i}.bias"])) for i in range(n + 1)]
        self.lut = t32(sam_norm_lut())
        self._tabs = {}

    # ---- preprocessing ------------------------------------------------------------------------------------
    def _tables(self, h, w):
        key = (h, w)
        if key not in self._tabs:
            nh, nw = resize_longest_side(h, w, self.cfg.image)
            dev = self.device

            def up(tab):
                b, k, ks = tab
                return (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks)

            th = up(resample.coeff_tables(w, nw, resample.BILINEAR)) if nw != w else None
            tv = up(resample.coeff_tables(h, nh, resample.BILINEAR)) if nh != h else None
            self._tabs[key] = (nh, nw, th, tv)
        return self._tabs[key]

    def preprocess(self, frames):
        """u8 [n,h,w,3] (channel order as handed over) -> (PIL-resized u8 [n,nh,nw,3], patch matrix f16 [n*g*g, 152])."""
        n, h, w, _ = frames.shape
        nh, nw, th, tv = self._tables(h, w)
        img = K.pil_resize(frames, nw, nh, th, tv, swap_rb=False)
        S = self.cfg.image
        return img, K.im2col_u8(img, self.lut, S, S, 7, 7, 4, 3, self.k_pad)

    # ---- network ------------------------------------------------------------------------------------------
    def trunk(self, patches, n):
        cfg = self.cfg
        g = self.grid0
        x = K.gemm(patches, self.pe_w, bias=self.pe_b, res=self.pos, res_rows=g * g, out_dtype=torch.float32)
        H = W = g
        stage_ends = set(int(v) for v in np.cumsum(cfg.blocks) - 1)
        stages, stages16 = [], []
        dev = x.device
        h_next = None
        for i, B in enumerate(self.blocks):
            dim, D, heads, qs = B["dim"], B["dim_out"], B["heads"], B["qs"]
            rows = n * H * W
            if self._attn8(i, H, W):  # [layer_norm1 ->] qkv -> window attention -> proj + residual in one launch (csrc/hiera.hip)
                if B["attn8_ln"]:
                    K.hiera_attn8(x, B["attn8"], n, H, W, heads, ln=(B["g1"], B["b1"], cfg.eps))
                else:
                    K.hiera_attn8(x, B["attn8"], n, H, W, heads, h=h_next if h_next is not None else K.layernorm(x, B["g1"], B["b1"], cfg.eps))
            elif "attnp" in B and K.hiera_attn_pool_ok(dim, D, heads, B["win"], H, W, qs):  # the stage-opening block: pooled q + shortcut
                x = K.hiera_attn_pool(h_next if h_next is not None else K.layernorm(x, B["g1"], B["b1"], cfg.eps), B["attnp"], n, H, W, heads, D)
                H, W = H // 2, W // 2
            elif "attn4" in B and K.hiera_attn4_ok(D, heads, B["win"], H, W, qs):  # the same for 4 x 4 windows at D = 224
                K.hiera_attn4(h_next if h_next is not None else K.layernorm(x, B["g1"], B["b1"], cfg.eps), x, B["attn4"], n, H, W, heads)
            else:
                # (a block whose predecessor ran the fused MLP gets its LayerNorm from that kernel: h_next)
                h = h_next if h_next is not None else K.layernorm(x, B["g1"], B["b1"], cfg.eps)
                if dim != D:
                    if qs and K.pooled_gemm_ok(rows, D):  # the shortcut's projection and its 2 x 2 max-pool in one launch
                        sc = K.gemm(h, B["wp"], bias=B["bp"], out_dtype=torch.float32, pool_hw=(H, W))
                    else:
                        sc = K.gemm(h, B["wp"], bias=B["bp"], out_dtype=torch.float32)
                        if qs:
                            pooled = torch.empty((n, H // 2, W // 2, D), dtype=torch.float32, device=dev)
                            K.maxpool2(sc.view(n, H, W, D), pooled)
                            sc = pooled.view(-1, D)
                    res = sc
                else:
                    res = x
                x, H, W = self._attention_half(B, h, x, res, n, H, W)
            h_next = None
            x16 = None
            if D in K.FUSED_MLP_WIDTHS and self.fused_mlp:
                if i in stage_ends:  # the FPN's lateral convolution reads this stage output as f16: written here, not cast later
                    x16 = torch.empty((n * H * W, D), dtype=torch.float16, device=dev)
                nxt = None
                if i + 1 < len(self.blocks) and not (self._attn8(i + 1, H, W) and self.blocks[i + 1]["attn8_ln"]):  # the next block's layer_norm1, on the rows while the kernel still holds them
                    h_next = torch.empty((n * H * W, D), dtype=torch.float16, device=dev)
                    nxt = (self.blocks[i + 1]["g1"], self.blocks[i + 1]["b1"], h_next)
                if "mlp_img" in B and K.ln_mlp_img_ok(D, n * H * W):  # the streamed-image form (csrc/hiera.hip): the next block's LayerNorm vectors are packed in
                    K.ln_mlp_img(x, B["mlp_img"], cfg.eps, x16=x16, h_next=h_next)
                else:
                    K.ln_mlp(x, B["g2"], B["b2"], B["w1"], B["bb1"], B["w2"], B["bb2"], cfg.eps, x16=x16, next_ln=nxt)  # one pass over x (csrc/mlp.hip)
            else:
                h2 = K.layernorm(x, B["g2"], B["b2"], cfg.eps)
                u = K.gemm(h2, B["w1"], bias=B["bb1"], act=K.ACT_GELU)
                K.gemm(u, B["w2"], bias=B["bb2"], res=x, out=x)
            if i in stage_ends:
                stages.append(x.view(n, H, W, D))
                stages16.append(x16)
                if i != len(self.blocks) - 1 and self.blocks[i + 1]["dim"] == self.blocks[i + 1]["dim_out"]:
                    x = x.clone()  # the stage output is kept; a same-width next block would update it in place
        return stages, stages16  # stages16: f16 copies of the stage outputs where the stage's last kernel wrote one (else None)

    def _attn8(self, i, H, W):
        """Block i runs its attention half (layer_norm1 included) in lmx_k_hiera_attn8."""
        B = self.blocks[i]
        return "attn8" in B and B["dim"] == B["dim_out"] and K.hiera_attn8_ok(B["dim_out"], B["heads"], B["win"], H, W, B["qs"])

    def _attention_half(self, B, h, x, res, n, H, W):
        """qkv GEMM -> [Q-pool] -> attention -> proj GEMM + residual as separate launches; returns (x, H, W) after the block's pooling."""
        D, heads, win, qs = B["dim_out"], B["heads"], B["win"], B["qs"]
        hd = D // heads
        rows = n * H * W
        dev = h.device
        Hq, Wq = H, W
        if qs and K.pooled_gemm_ok(rows, D):
            # pooled queries: the q third of the projection writes its 2 x 2 max-pool directly (a_mode 2), k and v come from
            # a second launch on the same rows — the full-resolution q is neither written nor read back by a pooling pass
            Hq, Wq = H // 2, W // 2
            q = K.gemm(h, B["wqkv"][:D], bias=B["bqkv"][:D], pool_hw=(H, W))
            kv = K.gemm(h, B["wqkv"][D:], bias=B["bqkv"][D:])
            k, v = kv[:, :D], kv[:, D:]
        else:
            qkv = K.gemm(h, B["wqkv"], bias=B["bqkv"])
            q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            if qs:
                Hq, Wq = H // 2, W // 2
                qp = torch.empty((n, Hq, Wq, D), dtype=torch.float16, device=dev)
                K.maxpool2(qkv.view(n, H, W, 3 * D)[..., :D], qp)
                q = qp.view(-1, D)
        a = torch.empty((n * Hq * Wq, D), dtype=torch.float16, device=dev)
        if win > 0:
            nW = -(-H // win) * -(-W // win)
            wq = win // 2 if qs else win
            K.attention(q, k, v, a, n * nW, heads, wq * wq, win * win, hd, hd ** -0.5,
                        window=dict(Gh=H, Gw=W, ws=win, q_stride=2 if qs else 1),
                        pad_k=B["padkv"][D:2 * D], pad_v=B["padkv"][2 * D:])
        else:
            K.attention(q, k, v, a, n, heads, Hq * Wq, H * W, hd, hd ** -0.5)
        xo = torch.empty((n * Hq * Wq, D), dtype=torch.float32, device=dev) if res is not x else x
        K.gemm(a, B["wo"], bias=B["bo"], res=res, out=xo)
        return xo, Hq, Wq

    def fpn(self, stages, stages16=None):
        cfg = self.cfg
        nlev = len(stages) - 1
        feats, prev = [], None
        for i in range(nlev, -1, -1):
            s = stages[i]
            n, H, W, C = s.shape
            a = stages16[i] if stages16 and stages16[i] is not None else K.cast_f16(s.view(-1, C))
            w, b = self.neck[i]
            out = torch.empty((n, H, W, cfg.fpn_dim), dtype=torch.float16, device=s.device)
            if i not in cfg.fpn_top_down or i == nlev:
                K.gemm(a, w, bias=b, out=out.view(-1, cfg.fpn_dim))
            else:
                up = torch.empty((n, H, W, cfg.fpn_dim), dtype=torch.float16, device=s.device)
                K.upsample2(prev, up)
                K.gemm(a, w, bias=b, res=up.view(-1, cfg.fpn_dim), out=out.view(-1, cfg.fpn_dim))
            prev = out
            feats.append(out)
        return feats[-3:][::-1]

    def encode_patches(self, patches, n):
        stages, stages16 = self.trunk(patches, n)
        return dict(fpn=self.fpn(stages, stages16), stages=stages)

    def encode(self, frames, precision=None):
        """(precision is accepted for interface parity with SamVitEncoder and ignored: the Hiera trunk has one plan, f16
        operands, which meets north_star's mask bar with margin — 0.9996 raw-frame IoU with the exact decoder.)"""
        img, patches = self.preprocess(frames)
        out = self.encode_patches(patches, frames.shape[0])
        out["resized"] = img
        return out


# ======================================================================================================================
# SAM v1 ImageEncoderViT — what `sam_model_registry["vit_b"|"vit_l"]` builds in services/sam3-pipeline/app/main.py:58-65
# (SURVEY.md Appendix A.2).  vit_h has head dim 80: the attention kernel's 96-wide head-dim class (csrc/attn.hip HDW).
# ======================================================================================================================
@dataclass
class SamVitConfig:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    global_idx: tuple = (2, 5, 8, 11)
    window: int = 14
    patch: int = 16
    image: int = 1024
    out_ch: int = 256
    eps: float = 1e-6

    @property
    def grid(self):
        return self.image // self.patch


def sam_vit_b():
    return SamVitConfig()


def sam_vit_l():
    return SamVitConfig(hidden=1024, layers=24, heads=16, mlp=4096, global_idx=(5, 11, 17, 23))


def sam_vit_h():
    return SamVitConfig(hidden=1280, layers=32, heads=16, mlp=5120, global_idx=(7, 15, 23, 31))


def vit_param_spec(cfg):
    """Ordered {transformers SamModel `vision_encoder.*` parameter name: (shape, init kind)}."""
    D, hd, g = cfg.hidden, cfg.hidden // cfg.heads, cfg.grid
    s = {"vision_encoder.pos_embed": ((1, g, g, D), "tok"),
         "vision_encoder.patch_embed.projection.weight": ((D, 3, cfg.patch, cfg.patch), "w"),
         "vision_encoder.patch_embed.projection.bias": ((D,), "b")}
    for i in range(cfg.layers):
        p = f"vision_encoder.layers.{i}."
        S = g if i in cfg.global_idx else cfg.window
        s[p + "layer_norm1.weight"] = ((D,), "g")
        s[p + "layer_norm1.bias"] = ((D,), "b")
        s[p + "attn.rel_pos_h"] = ((2 * S - 1, hd), "b")
        s[p + "attn.rel_pos_w"] = ((2 * S - 1, hd), "b")
        s[p + "attn.qkv.weight"] = ((3 * D, D), "w")
        s[p + "attn.qkv.bias"] = ((3 * D,), "b")
        s[p + "attn.proj.weight"] = ((D, D), "w")
        s[p + "attn.proj.bias"] = ((D,), "b")
        s[p + "layer_norm2.weight"] = ((D,), "g")
        s[p + "layer_norm2.bias"] = ((D,), "b")
        s[p + "mlp.lin1.weight"] = ((cfg.mlp, D), "w")
        s[p + "mlp.lin1.bias"] = ((cfg.mlp,), "b")
        s[p + "mlp.lin2.weight"] = ((D, cfg.mlp), "w")
        s[p + "mlp.lin2.bias"] = ((D,), "b")
    s["vision_encoder.neck.conv1.weight"] = ((cfg.out_ch, D, 1, 1), "w")
    s["vision_encoder.neck.layer_norm1.weight"] = ((cfg.out_ch,), "g")
    s["vision_encoder.neck.layer_norm1.bias"] = ((cfg.out_ch,), "b")
    s["vision_encoder.neck.conv2.weight"] = ((cfg.out_ch, cfg.out_ch, 3, 3), "w")
    s["vision_encoder.neck.layer_norm2.weight"] = ((cfg.out_ch,), "g")
    s["vision_encoder.neck.layer_norm2.bias"] = ((cfg.out_ch,), "b")
    return s


class SamVitEncoder:
    """Device-resident SAM v1 image encoder: patch-embed GEMM (+abs pos), windowed (14x14, zero-padded 64->70) and global
    attention with decomposed relative-position bias, MLP, neck (1x1 -> LN2d -> 3x3 -> LN2d).  ``encode(frames)`` ->
    dict(fpn=[None, None, embedding f16 [n,64,64,256]], resized=...) — same shape contract as HieraEncoder for the decoder."""

    def __init__(self, cfg, state_dict, device="cuda", precision="exact"):
        """precision: the default plan of encode() — "exact" (services, adapters: every weight as the two-term f16 split
        [whi | wlo], one launch per Linear with lmx_k_gemm's a_rep = 2; f16 weights alone carry 3.9e-4 of the path's 9e-4
        relative logit error, profiles/r03_sam_vit_precision_probe.txt) or "f16" (throughput: 2x less MFMA work)."""
        if precision not in ("exact", "f16"):
            raise ValueError(f"precision {precision!r}: expected 'exact' or 'f16'")
        self.precision = precision
        self.cfg = cfg
        self.device = torch.device(device)
        dev = self.device
        sd = state_dict
        D, P = cfg.hidden, cfg.patch
        self._w32, self._w2 = {}, {}  # f32 [N, K] weights by key, and their [whi | wlo] f16 [N, 2K] form (built on first exact use)

        def t32(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        def t16(a, key=None):
            if key is not None:
                self._w32[key] = np.ascontiguousarray(a, dtype=np.float32)
            return t32(a).to(torch.float16).contiguous()

        w = np.transpose(sd["vision_encoder.patch_embed.projection.weight"], (0, 2, 3, 1)).reshape(D, P * P * 3)
        self._w32["pe"] = np.ascontiguousarray(w, dtype=np.float32)
        self.k_pad = P * P * 3  # 768: already a multiple of 8
        self.pe_w, self.pe_b = t16(w), t32(sd["vision_encoder.patch_embed.projection.bias"])
        self.pos = t32(sd["vision_encoder.pos_embed"].reshape(cfg.grid * cfg.grid, D))
        self.layers = []
        for i in range(cfg.layers):
            p = f"vision_encoder.layers.{i}."
            qb = sd[p + "attn.qkv.bias"]
            self.layers.append(dict(
                glob=i in cfg.global_idx,
                g1=t32(sd[p + "layer_norm1.weight"]), b1=t32(sd[p + "layer_norm1.bias"]),
                wqkv=t16(sd[p + "attn.qkv.weight"], f"{i}.qkv"), bqkv=t32(qb), padkv=t16(qb),
                rh=t32(sd[p + "attn.rel_pos_h"]), rw=t32(sd[p + "attn.rel_pos_w"]),
                wo=t16(sd[p + "attn.proj.weight"], f"{i}.proj"), bo=t32(sd[p + "attn.proj.bias"]),
                g2=t32(sd[p + "layer_norm2.weight"]), b2=t32(sd[p + "layer_norm2.bias"]),
                w1=t16(sd[p + "mlp.lin1.weight"], f"{i}.fc1"), bb1=t32(sd[p + "mlp.lin1.bias"]),
                w2=t16(sd[p + "mlp.lin2.weight"], f"{i}.fc2"), bb2=t32(sd[p + "mlp.lin2.bias"])))
        self.n1_w = t16(sd["vision_encoder.neck.conv1.weight"][:, :, 0, 0], "n1")
        self.n1_ln = (t32(sd["vision_encoder.neck.layer_norm1.weight"]), t32(sd["vision_encoder.neck.layer_norm1.bias"]))
        c2 = sd["vision_encoder.neck.conv2.weight"]
        self.n2_w = t16(np.transpose(c2, (0, 2, 3, 1)).reshape(c2.shape[0], -1), "n2")
        self.n2_ln = (t32(sd["vision_encoder.neck.layer_norm2.weight"]), t32(sd["vision_encoder.neck.layer_norm2.bias"]))
        self.lut = t32(sam_norm_lut())
        self._tabs = {}

    _tables = HieraEncoder._tables

    def preprocess(self, frames):
        n, h, w, _ = frames.shape
        nh, nw, th, tv = self._tables(h, w)
        img = K.pil_resize(frames, nw, nh, th, tv, swap_rb=False)
        S, P = self.cfg.image, self.cfg.patch
        return img, K.im2col_u8(img, self.lut, S, S, P, P, P, 0, self.k_pad)

    def _split2(self, key):
        """[whi | wlo] f16 [N, 2K] of the f32 weight `key`: whi = f16(w), wlo = f16(w - whi) (often subnormal: the f16 MFMA honours
        subnormal operands on gfx950, tools/mfma_denorm_probe.py) — w to 2^-22 relative, or 3e-8 absolute for tiny weights."""
        if key not in self._w2:
            w = self._w32[key]
            hi = w.astype(np.float16)
            lo = (w - hi.astype(np.float32)).astype(np.float16)
            self._w2[key] = torch.from_numpy(np.ascontiguousarray(np.concatenate([hi, lo], 1))).to(self.device)
        return self._w2[key]

    def embed(self, patches, n, precision=None):
        cfg = self.cfg
        precision = precision or self.precision
        if precision not in ("exact", "f16"):
            raise ValueError(f"precision {precision!r}: expected 'exact' or 'f16'")
        exact = precision == "exact"
        g, D, H = cfg.grid, cfg.hidden, cfg.heads
        hd = D // H
        rows = n * g * g

        def lin(a, w16, key, **kw):  # one launch either way: f16 weights, or a_rep = 2 over [whi | wlo]
            if exact:
                w2 = self._split2(key)
                if a.shape[1] % 64 == 0 and a.shape[0] >= 512 and w2.shape[0] >= 96 and w2.shape[0] % 8 == 0:
                    return K.gemm(a, w2, a_rep=2, **kw)
                # shapes outside the LDS-DMA kernel (toy configurations; ViT-B / L / H never get here): the same sum over an
                # explicit [a | a] copy
                return K.gemm(torch.cat([a, a], 1), w2, **kw)
            return K.gemm(a, w16, **kw)

        x = lin(patches, self.pe_w, "pe", bias=self.pe_b, res=self.pos, res_rows=g * g, out_dtype=torch.float32)
        for i, L in enumerate(self.layers):
            h = K.layernorm(x, L["g1"], L["b1"], cfg.eps)
            qkv = lin(h, L["wqkv"], f"{i}.qkv", bias=L["bqkv"])
            q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            a = torch.empty((rows, D), dtype=torch.float16, device=x.device)
            if L["glob"]:
                K.attention(q, k, v, a, n, H, g * g, g * g, hd, hd ** -0.5, rel_pos=(L["rh"], L["rw"]))
            else:
                ws = cfg.window
                nW = (-(-g // ws)) ** 2
                K.attention(q, k, v, a, n * nW, H, ws * ws, ws * ws, hd, hd ** -0.5, window=dict(Gh=g, Gw=g, ws=ws, q_stride=1),
                            pad_k=L["padkv"][D:2 * D], pad_v=L["padkv"][2 * D:], rel_pos=(L["rh"], L["rw"]))
            lin(a, L["wo"], f"{i}.proj", bias=L["bo"], res=x, out=x)
            h2 = K.layernorm(x, L["g2"], L["b2"], cfg.eps)
            u = lin(h2, L["w1"], f"{i}.fc1", bias=L["bb1"], act=K.ACT_GELU)
            lin(u, L["w2"], f"{i}.fc2", bias=L["bb2"], res=x, out=x)
        # neck: 1x1 (no bias) -> LayerNorm2d -> 3x3 (no bias) -> LayerNorm2d
        y = lin(K.cast_f16(x), self.n1_w, "n1", out_dtype=torch.float32)
        y = K.layernorm(y, *self.n1_ln, 1e-6)
        y4 = y.view(n, g, g, cfg.out_ch)
        if exact:  # the 3x3 as two accumulating launches of the implicit GEMM (whi, then + wlo), f32 output, f32 embedding
            if "n2.hi" not in self._w2:
                w2 = self._split2("n2")
                Kc = w2.shape[1] // 2
                self._w2["n2.hi"], self._w2["n2.lo"] = w2[:, :Kc].contiguous(), w2[:, Kc:].contiguous()
            acc = K.conv3x3(y4, self._w2["n2.hi"], bias=None, act=K.ACT_NONE, out_dtype=torch.float32)
            acc = K.conv3x3(y4, self._w2["n2.lo"], bias=None, act=K.ACT_NONE, res=acc, out=acc)
            y = K.layernorm(acc.view(rows, cfg.out_ch), *self.n2_ln, 1e-6, out_dtype=torch.float32)
        else:
            y = K.conv3x3(y4, self.n2_w, bias=None, act=K.ACT_NONE)
            y = K.layernorm(y.view(rows, cfg.out_ch), *self.n2_ln, 1e-6)
        return y.view(n, g, g, cfg.out_ch)

    def encode(self, frames, precision=None):
        img, patches = self.preprocess(frames)
        return dict(fpn=[None, None, self.embed(patches, frames.shape[0], precision)], resized=img)
