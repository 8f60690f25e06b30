"""oracle.hiera — fp32 CPU restatement of the SAM2 Hiera image encoder (trunk + FPN neck), the BASELINE cfg#3
architecture ("SAM3 Hiera-B+ image encoder").  TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference's sam3 service calls segment_anything (SAM v1), which is not installed; the Hiera trunk is the encoder
BASELINE.json names.  Pinned against transformers' Sam2VisionModel (tests/test_oracle_hiera.py), following
  TF:models/sam2/modeling_sam2.py:106-136 (patch embed), :616-644 (windowed pos embed), :457-546 (multi-scale block),
  :291-364 (Q-pooled attention), :412-455 (window partition with padding), :216-265 (FPN neck), :142-200 (sine pos).
State-dict names are transformers' (backbone.*, neck.*)."""
import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


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
        """[(dim_in, dim_out, heads, window, q_stride)] per block, as Sam2MultiScaleBlock.__init__ derives them."""
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


def hiera_tiny_test():
    """Small config with every code path: Q-pool at 3 stage changes, padded windows (16 % 7 != 0 ...), a global block."""
    return HieraConfig(hidden=16, blocks=(1, 2, 3, 2), dims=(16, 32, 64, 128), heads=(1, 2, 4, 8), windows=(8, 4, 14, 7),
                       global_blocks=(4,), pos_bkg=(7, 7), fpn_dim=32, image=256)


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def _window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)
    return x, (Hp, Wp)


def _window_unpartition(w, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // ((Hp // ws) * (Wp // ws))
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def _pool(x):  # [B,H,W,C] 2x2 max
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def pos_embed(cfg, sd, h, w):
    pe = F.interpolate(_t(sd, "backbone.pos_embed"), size=(h, w), mode="bicubic")
    win = _t(sd, "backbone.pos_embed_window")
    pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
    return pe.permute(0, 2, 3, 1)


def trunk_forward(cfg, sd, pixel_values, return_blocks=False):
    """pixel_values f32 [B,3,H,W] -> list of the 4 stage outputs [B,h,w,dim] (NHWC, like the HF model)."""
    x = F.conv2d(pixel_values, _t(sd, "backbone.patch_embed.projection.weight"),
                 _t(sd, "backbone.patch_embed.projection.bias"), stride=4, padding=3).permute(0, 2, 3, 1)
    x = x + pos_embed(cfg, sd, x.shape[1], x.shape[2])
    stage_ends = [sum(cfg.blocks[:i + 1]) - 1 for i in range(len(cfg.blocks))]
    outs, per_block = [], []
    for i, (dim, dim_out, heads, win, qs) in enumerate(cfg.block_plan()):
        p = f"backbone.blocks.{i}."
        residual = x
        h = F.layer_norm(x, (dim,), _t(sd, p + "layer_norm1.weight"), _t(sd, p + "layer_norm1.bias"), cfg.eps)
        if dim != dim_out:
            residual = F.linear(h, _t(sd, p + "proj.weight"), _t(sd, p + "proj.bias"))
            residual = _pool(residual) if qs else residual
        H, W = h.shape[1], h.shape[2]
        ws = win
        if win > 0:
            h, pad_hw = _window_partition(h, win)
        B, hh, ww, _ = h.shape
        qkv = F.linear(h, _t(sd, p + "attn.qkv.weight"), _t(sd, p + "attn.qkv.bias")).reshape(B, hh * ww, 3, heads, -1)
        q, k, v = torch.unbind(qkv, 2)
        if qs:
            q = _pool(q.reshape(B, hh, ww, -1))
            hh, ww = q.shape[1:3]
            q = q.reshape(B, hh * ww, heads, -1)
        hd = dim_out // heads
        aw = torch.matmul(q.transpose(1, 2), k.transpose(1, 2).transpose(2, 3)) * hd ** -0.5
        aw = F.softmax(aw, dim=-1, dtype=torch.float32)
        a = torch.matmul(aw, v.transpose(1, 2)).transpose(1, 2).reshape(B, hh, ww, -1)
        a = F.linear(a, _t(sd, p + "attn.proj.weight"), _t(sd, p + "attn.proj.bias"))
        if qs:
            ws = win // 2
            H, W = residual.shape[1:3]
            pad_hw = (H + (-H) % ws, W + (-W) % ws) if win > 0 else None
        if win > 0:
            a = _window_unpartition(a, ws, pad_hw, (H, W))
        x = residual + a
        h2 = F.layer_norm(x, (dim_out,), _t(sd, p + "layer_norm2.weight"), _t(sd, p + "layer_norm2.bias"), cfg.eps)
        h2 = F.linear(F.gelu(F.linear(h2, _t(sd, p + "mlp.proj_in.weight"), _t(sd, p + "mlp.proj_in.bias"))),
                      _t(sd, p + "mlp.proj_out.weight"), _t(sd, p + "mlp.proj_out.bias"))
        x = x + h2
        per_block.append(x)
        if i in stage_ends:
            outs.append(x)
    return (outs, per_block) if return_blocks else outs


def sine_pos(fpn_dim, h, w):
    """Sam2SinePositionEmbedding(num_position_features=fpn_dim//2, normalize=True) for a [1,*,h,w] map -> [h,w,fpn_dim]."""
    npf = fpn_dim // 2
    y = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32)[None, :].expand(h, w)
    y = y / (y[-1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(npf, dtype=torch.int64).to(torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2)


def neck_forward(cfg, sd, stage_outs):
    """-> (fpn features high->low resolution, 3 levels NCHW [B,fpn_dim,h,w]) like Sam2VisionModel.fpn_hidden_states."""
    n = len(stage_outs) - 1
    feats, prev = [], None
    for i in range(n, -1, -1):
        lat = F.conv2d(stage_outs[i].permute(0, 3, 1, 2), _t(sd, f"neck.convs.{n - i}.weight"), _t(sd, f"neck.convs.{n - i}.bias"))
        if i not in cfg.fpn_top_down or i == n:
            prev = lat
        else:
            prev = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        feats.append(prev)
    return feats[-3:][::-1]


def encoder_forward(cfg, sd, pixel_values):
    outs = trunk_forward(cfg, sd, pixel_values)
    return neck_forward(cfg, sd, outs), outs
