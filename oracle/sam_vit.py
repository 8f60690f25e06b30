"""oracle.sam_vit — fp32 CPU restatement of SAM v1's ImageEncoderViT (what `set_image` runs in
services/sam3-pipeline/app/main.py:80 when a `sam_vit_*.pth` checkpoint is present).  TEST INFRASTRUCTURE (see
oracle/__init__.py).  Pinned against transformers' SamModel.vision_encoder (tests/test_oracle_sam_vit.py), following
TF:models/sam/modeling_sam.py:1019-1066 (patch embed + abs pos), :700-835 (attention with decomposed rel-pos),
:891-972 (layer with window partition / zero padding), :975-992 (neck)."""
import torch
import torch.nn.functional as F


EMULATE = set()  # f16-STORAGE emulation switches (tools/sam_precision_probe.py): which of the device path's f16 roundings to
#                  apply to this fp32 arithmetic.  Empty = the plain fp32 oracle (the parity target).  Tags: "w" weights, "ln"
#                  LayerNorm outputs, "qkv", "rel" (decomposed rel-pos tables), "p" (softmax probabilities), "ao" (attention
#                  output), "gelu" (MLP hidden), "neck" (the neck's three f16 tensors), "emb" (the f16 image embedding)


def _q(x, tag):
    return x.half().float() if tag in EMULATE else x


def _t(sd, k):
    v = sd[k]
    v = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
    return _q(v, "w") if (k.endswith("weight") and v.dim() >= 2) else v


def _rel(q_size, rel_pos):
    idx = (torch.arange(q_size)[:, None] - torch.arange(q_size)[None, :]) + (q_size - 1)
    return rel_pos[idx.long()]  # [q, k, c]; sizes are equal, so get_rel_pos does not interpolate


def _attention(sd, p, x, heads):
    B, H, W, D = x.shape
    hd = D // heads
    qkv = _q(F.linear(x, _t(sd, p + "qkv.weight"), _t(sd, p + "qkv.bias")), "qkv").reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, hd).unbind(0)
    a = (q * hd ** -0.5) @ k.transpose(-2, -1)
    rq = q.reshape(B * heads, H, W, hd)
    rel_h = _q(torch.einsum("bhwc,hkc->bhwk", rq, _rel(H, _t(sd, p + "rel_pos_h"))), "rel")
    rel_w = _q(torch.einsum("bhwc,wkc->bhwk", rq, _rel(W, _t(sd, p + "rel_pos_w"))), "rel")
    a = a + (rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).reshape_as(a)
    a = _q(torch.softmax(a, dim=-1), "p")
    o = _q((a @ v).reshape(B, heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, D), "ao")
    return F.linear(o, _t(sd, p + "proj.weight"), _t(sd, p + "proj.bias"))


def encoder_forward(cfg, sd, pixel_values):
    """pixel_values f32 [B,3,1024,1024] -> image embedding f32 [B,256,64,64]."""
    D = cfg.hidden
    x = F.conv2d(pixel_values, _t(sd, "vision_encoder.patch_embed.projection.weight"),
                 _t(sd, "vision_encoder.patch_embed.projection.bias"), stride=cfg.patch).permute(0, 2, 3, 1)
    x = x + _t(sd, "vision_encoder.pos_embed")
    for i in range(cfg.layers):
        p = f"vision_encoder.layers.{i}."
        res = x
        h = _q(F.layer_norm(x, (D,), _t(sd, p + "layer_norm1.weight"), _t(sd, p + "layer_norm1.bias"), cfg.eps), "ln")
        if i in cfg.global_idx:
            h = _attention(sd, p + "attn.", h, cfg.heads)
        else:
            ws = cfg.window
            B, H, W, _ = h.shape
            ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
            hp = F.pad(h, (0, 0, 0, pw, 0, ph))
            Hp, Wp = H + ph, W + pw
            win = hp.reshape(B, Hp // ws, ws, Wp // ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, D)
            win = _attention(sd, p + "attn.", win, cfg.heads)
            h = win.reshape(B, Hp // ws, Wp // ws, ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)[:, :H, :W]
        x = res + h
        h = _q(F.layer_norm(x, (D,), _t(sd, p + "layer_norm2.weight"), _t(sd, p + "layer_norm2.bias"), cfg.eps), "ln")
        h = F.linear(_q(F.gelu(F.linear(h, _t(sd, p + "mlp.lin1.weight"), _t(sd, p + "mlp.lin1.bias"))), "gelu"),
                     _t(sd, p + "mlp.lin2.weight"), _t(sd, p + "mlp.lin2.bias"))
        x = x + h
    y = F.conv2d(_q(x, "neck").permute(0, 3, 1, 2), _t(sd, "vision_encoder.neck.conv1.weight"))
    y = _q(F.layer_norm(y.permute(0, 2, 3, 1), (cfg.out_ch,), _t(sd, "vision_encoder.neck.layer_norm1.weight"),
                        _t(sd, "vision_encoder.neck.layer_norm1.bias"), 1e-6), "neck").permute(0, 3, 1, 2)
    y = _q(F.conv2d(y, _t(sd, "vision_encoder.neck.conv2.weight"), padding=1), "neck")
    y = F.layer_norm(y.permute(0, 2, 3, 1), (cfg.out_ch,), _t(sd, "vision_encoder.neck.layer_norm2.weight"),
                     _t(sd, "vision_encoder.neck.layer_norm2.bias"), 1e-6).permute(0, 3, 1, 2)
    return _q(y, "emb")
