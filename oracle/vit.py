"""oracle.vit — fp32 CPU restatement of the DINO ViT forwards the dinov3 service calls
(services/dinov3-pipeline/app/main.py:110-113).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows (third-party code, not in /root/reference; transformers 5.15.0 as installed):
  DINOv3: TF:models/dinov3_vit/modeling_dinov3_vit.py:60-92 (embeddings), :153-200 (RoPE tables), :238-268
          (apply_rotary_pos_emb), :275-330 (attention), :400-445 (layer), :507-545 (model: final LayerNorm)
  DINOv2: TF:models/dinov2/modeling_dinov2.py:57-149 (embeddings + pos interpolation), :199-260 (attention),
          :342-380 (layer), :433-480 (model)
State dicts use the transformers parameter names (lmx.dino.param_spec)."""
import math

import torch
import torch.nn.functional as F


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def rope_tables(head_dim, theta, gh, gw):
    inv_freq = 1 / theta ** torch.arange(0, 1, 4 / head_dim, dtype=torch.float32)
    ch = torch.arange(0.5, gh, dtype=torch.float32) / gh
    cw = torch.arange(0.5, gw, dtype=torch.float32) / gw
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    ang = 2 * math.pi * coords[:, :, None] * inv_freq[None, None, :]
    ang = ang.flatten(1, 2).tile(2)
    return torch.cos(ang), torch.sin(ang)


def _attention(q, k, v, heads):
    B, T, D = q.shape
    hd = D // heads
    q, k, v = (t.view(B, T, heads, hd).transpose(1, 2) for t in (q, k, v))
    w = torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5
    w = F.softmax(w, dim=-1)
    return torch.matmul(w, v).transpose(1, 2).reshape(B, T, D)


def dinov3_forward(cfg, sd, pixel_values, return_layers=False):
    """pixel_values f32 [B,3,H,W] -> last_hidden_state f32 [B, 1+R+np, D]."""
    D, heads, P = cfg.hidden, cfg.heads, cfg.patch
    B, _, H, W = pixel_values.shape
    x = F.conv2d(pixel_values, _t(sd, "embeddings.patch_embeddings.weight"), _t(sd, "embeddings.patch_embeddings.bias"),
                 stride=P)
    x = x.flatten(2).transpose(1, 2)
    cls = _t(sd, "embeddings.cls_token").expand(B, -1, -1)
    reg = _t(sd, "embeddings.register_tokens").expand(B, -1, -1)
    x = torch.cat([cls, reg, x], dim=1)
    npre = 1 + reg.shape[1]
    cos, sin = rope_tables(D // heads, cfg.rope_theta, H // P, W // P)
    outs = []
    for i in range(cfg.layers):
        p = f"model.layer.{i}."
        h = F.layer_norm(x, (D,), _t(sd, p + "norm1.weight"), _t(sd, p + "norm1.bias"), cfg.eps)
        q = F.linear(h, _t(sd, p + "attention.q_proj.weight"), _t(sd, p + "attention.q_proj.bias"))
        kb = _t(sd, p + "attention.k_proj.bias") if (p + "attention.k_proj.bias") in sd else None
        k = F.linear(h, _t(sd, p + "attention.k_proj.weight"), kb)
        v = F.linear(h, _t(sd, p + "attention.v_proj.weight"), _t(sd, p + "attention.v_proj.bias"))
        T = q.shape[1]
        hd = D // heads

        def rope(t):
            t = t.view(B, T, heads, hd).transpose(1, 2)
            pre, pat = t[..., :npre, :], t[..., npre:, :]
            pat = pat * cos + _rotate_half(pat) * sin
            return torch.cat((pre, pat), dim=-2).transpose(1, 2).reshape(B, T, D)

        a = _attention(rope(q), rope(k), v, heads)
        a = F.linear(a, _t(sd, p + "attention.o_proj.weight"), _t(sd, p + "attention.o_proj.bias"))
        x = a * _t(sd, p + "layer_scale1.lambda1") + x
        h = F.layer_norm(x, (D,), _t(sd, p + "norm2.weight"), _t(sd, p + "norm2.bias"), cfg.eps)
        h = F.linear(h, _t(sd, p + "mlp.up_proj.weight"), _t(sd, p + "mlp.up_proj.bias"))
        h = F.linear(F.gelu(h), _t(sd, p + "mlp.down_proj.weight"), _t(sd, p + "mlp.down_proj.bias"))
        x = h * _t(sd, p + "layer_scale2.lambda1") + x
        if return_layers:
            outs.append(x)
    y = F.layer_norm(x, (D,), _t(sd, "norm.weight"), _t(sd, "norm.bias"), cfg.eps)
    return (y, outs) if return_layers else y


def dinov2_forward(cfg, sd, pixel_values):
    """pixel_values f32 [B,3,H,W] -> last_hidden_state f32 [B, 1+np, D]."""
    D, heads, P = cfg.hidden, cfg.heads, cfg.patch
    B, _, H, W = pixel_values.shape
    x = F.conv2d(pixel_values, _t(sd, "embeddings.patch_embeddings.projection.weight"),
                 _t(sd, "embeddings.patch_embeddings.projection.bias"), stride=P)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([_t(sd, "embeddings.cls_token").expand(B, -1, -1), x], dim=1)
    pos = _t(sd, "embeddings.position_embeddings")
    gi = int(round((pos.shape[1] - 1) ** 0.5))
    gh, gw = H // P, W // P
    if (gi, gi) != (gh, gw):
        pp = pos[:, 1:].reshape(1, gi, gi, D).permute(0, 3, 1, 2)
        pp = F.interpolate(pp.to(torch.float32), size=(gh, gw), mode="bicubic", align_corners=False)
        pos = torch.cat((pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, D)), dim=1)
    x = x + pos
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        h = F.layer_norm(x, (D,), _t(sd, p + "norm1.weight"), _t(sd, p + "norm1.bias"), cfg.eps)
        q, k, v = (F.linear(h, _t(sd, p + f"attention.attention.{n}.weight"), _t(sd, p + f"attention.attention.{n}.bias"))
                   for n in ("query", "key", "value"))
        a = _attention(q, k, v, heads)
        a = F.linear(a, _t(sd, p + "attention.output.dense.weight"), _t(sd, p + "attention.output.dense.bias"))
        x = a * _t(sd, p + "layer_scale1.lambda1") + x
        h = F.layer_norm(x, (D,), _t(sd, p + "norm2.weight"), _t(sd, p + "norm2.bias"), cfg.eps)
        h = F.linear(h, _t(sd, p + "mlp.fc1.weight"), _t(sd, p + "mlp.fc1.bias"))
        h = F.linear(F.gelu(h), _t(sd, p + "mlp.fc2.weight"), _t(sd, p + "mlp.fc2.bias"))
        x = h * _t(sd, p + "layer_scale2.lambda1") + x
    return F.layer_norm(x, (D,), _t(sd, "layernorm.weight"), _t(sd, "layernorm.bias"), cfg.eps)


def dino_forward(cfg, sd, pixel_values):
    return dinov3_forward(cfg, sd, pixel_values) if cfg.arch == "dinov3" else dinov2_forward(cfg, sd, pixel_values)


def embed(cfg, sd, pixel_values):
    """The service's pooling: mean over ALL tokens (services/dinov3-pipeline/app/main.py:113)."""
    return dino_forward(cfg, sd, pixel_values).mean(dim=1)
