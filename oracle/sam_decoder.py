"""oracle.sam_decoder — fp32 CPU restatement of the box-prompted mask path of ``SamPredictor.predict(box=..., multimask_output=False)``
(services/sam3-pipeline/app/main.py:83-89): prompt encoder, two-way-transformer mask decoder, and SamPredictor's
post-processing to a boolean mask at the frame's resolution.  TEST INFRASTRUCTURE (see oracle/__init__.py).

segment_anything is not installed; the architecture-equivalent implementation in transformers (SamModel) pins this file
(tests/test_oracle_sam_decoder.py): TF:models/sam/modeling_sam.py:596-698 (prompt encoder), :205-268 (attention),
:271-345 (two-way block), :348-406 (two-way transformer), :432-543 (mask decoder), :1128-1139 (image-wide positional
embedding).  Post-processing follows segment_anything's Sam.postprocess_masks (public source): bilinear to 1024^2
(align_corners=False), crop to the resized frame, bilinear to the original size, > mask_threshold (0.0).
State-dict names are transformers' SamModel names (prompt_encoder.*, mask_decoder.*, shared_image_embedding.*).
The decoder only needs a [n,256,64,64] image embedding, so it serves both encoders (Hiera FPN level 2 and SAM ViT neck)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

HEADS = 8


EMULATE = set()  # f16-STORAGE emulation switches (tools/sam_precision_probe.py), as in oracle/sam_vit.py: "w" weights, "proj"
#                  (inputs and outputs of the attention projections, attention output), "mlp" (token MLP input / hidden),
#                  "up" (upscaler: keys16, LN+GELU output, second ConvTranspose output), "tok" (mask / iou token, hypernet hidden)


def _q(x, tag):
    return x.half().float() if tag in EMULATE else x


def _t(sd, k):
    v = sd[k]
    v = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
    return _q(v, "w") if (k.endswith("weight") and v.dim() >= 2 and "embed" not in k and "token" not in k) else v


def _pe(gauss, coords01):
    """SamPositionalEmbedding.forward for coordinates already normalised to [0,1]."""
    c = 2 * coords01 - 1
    c = c @ gauss
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def image_pe(sd, size=64):
    grid = torch.ones((size, size), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / size
    x = (grid.cumsum(dim=1) - 0.5) / size
    return _pe(_t(sd, "shared_image_embedding.positional_embedding"), torch.stack([x, y], dim=-1))  # [64,64,256]


def prompt_encode_box(sd, boxes, image_size=1024):
    """boxes f32 [n,4] xyxy in the resized-image (1024-space) frame -> sparse [n,2,256]."""
    b = boxes.to(torch.float32) + 0.5
    coords = b.reshape(-1, 2, 2).clone()
    coords[..., 0] = coords[..., 0] / image_size
    coords[..., 1] = coords[..., 1] / image_size
    emb = _pe(_t(sd, "shared_image_embedding.positional_embedding"), coords)  # one tied Fourier matrix
    emb[:, 0, :] += _t(sd, "prompt_encoder.point_embed.2.weight")[0]
    emb[:, 1, :] += _t(sd, "prompt_encoder.point_embed.3.weight")[0]
    return emb


def _attn(sd, p, q, k, v):
    q = _q(F.linear(_q(q, "proj"), _t(sd, p + "q_proj.weight"), _t(sd, p + "q_proj.bias")), "proj")
    k = _q(F.linear(_q(k, "proj"), _t(sd, p + "k_proj.weight"), _t(sd, p + "k_proj.bias")), "proj")
    v = _q(F.linear(_q(v, "proj"), _t(sd, p + "v_proj.weight"), _t(sd, p + "v_proj.bias")), "proj")
    n, tq, c = q.shape
    hd = c // HEADS
    qh, kh, vh = (t.reshape(n, -1, HEADS, hd).transpose(1, 2) for t in (q, k, v))
    w = _q(torch.softmax(torch.matmul(qh, kh.transpose(2, 3)) * hd ** -0.5, dim=-1), "proj")
    o = _q(torch.matmul(w, vh).transpose(1, 2).reshape(n, tq, c), "proj")
    return F.linear(o, _t(sd, p + "out_proj.weight"), _t(sd, p + "out_proj.bias"))


def _ln(sd, p, x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), _t(sd, p + "weight"), _t(sd, p + "bias"), eps)


def _ffn(sd, p, x, n_mid):
    x = _q(F.relu(F.linear(_q(x, "tok"), _t(sd, p + "proj_in.weight"), _t(sd, p + "proj_in.bias"))), "tok")
    for i in range(n_mid):
        x = _q(F.relu(F.linear(x, _t(sd, p + f"layers.{i}.weight"), _t(sd, p + f"layers.{i}.bias"))), "tok")
    return F.linear(x, _t(sd, p + "proj_out.weight"), _t(sd, p + "proj_out.bias"))


def mask_decode(sd, image_emb, sparse, return_all=False):
    """image_emb f32 [n,256,64,64], sparse [n,2,256] -> (low-res logits of mask 0 [n,256,256], iou [n])."""
    n, c, h, w = image_emb.shape
    tokens = torch.cat([_t(sd, "mask_decoder.iou_token.weight"), _t(sd, "mask_decoder.mask_tokens.weight")], 0)
    tokens = torch.cat([tokens[None].expand(n, -1, -1), sparse], dim=1)  # [n,7,256]
    dense = _t(sd, "prompt_encoder.no_mask_embed.weight").reshape(1, -1, 1, 1)
    keys = (image_emb + dense).flatten(2).transpose(1, 2)  # [n,4096,256]
    key_pe = image_pe(sd, h).reshape(1, h * w, c)
    queries, qpe = tokens, tokens
    for i in range(2):
        p = f"mask_decoder.transformer.layers.{i}."
        if i == 0:
            queries = _attn(sd, p + "self_attn.", queries, queries, queries)
        else:
            q = queries + qpe
            queries = queries + _attn(sd, p + "self_attn.", q, q, queries)
        queries = _ln(sd, p + "layer_norm1.", queries)
        queries = queries + _attn(sd, p + "cross_attn_token_to_image.", queries + qpe, keys + key_pe, keys)
        queries = _ln(sd, p + "layer_norm2.", queries)
        m = F.linear(_q(F.relu(F.linear(_q(queries, "mlp"), _t(sd, p + "mlp.lin1.weight"), _t(sd, p + "mlp.lin1.bias"))), "mlp"),
                     _t(sd, p + "mlp.lin2.weight"), _t(sd, p + "mlp.lin2.bias"))
        queries = _ln(sd, p + "layer_norm3.", queries + m)
        keys = keys + _attn(sd, p + "cross_attn_image_to_token.", keys + key_pe, queries + qpe, queries)
        keys = _ln(sd, p + "layer_norm4.", keys)
    p = "mask_decoder.transformer."
    queries = queries + _attn(sd, p + "final_attn_token_to_image.", queries + qpe, keys + key_pe, keys)
    queries = F.layer_norm(queries, (c,), _t(sd, p + "layer_norm_final_attn.weight"), _t(sd, p + "layer_norm_final_attn.bias"), 1e-5)
    iou_tok, mask_tok = queries[:, 0], queries[:, 1:5]
    x = _q(keys, "up").transpose(1, 2).reshape(n, c, h, w)
    x = F.conv_transpose2d(x, _t(sd, "mask_decoder.upscale_conv1.weight"), _t(sd, "mask_decoder.upscale_conv1.bias"), stride=2)
    x = _q(F.gelu(_ln(sd, "mask_decoder.upscale_layer_norm.", x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)), "up")
    x = _q(F.gelu(F.conv_transpose2d(x, _t(sd, "mask_decoder.upscale_conv2.weight"), _t(sd, "mask_decoder.upscale_conv2.bias"), stride=2)), "up")
    hyper = torch.stack([_ffn(sd, f"mask_decoder.output_hypernetworks_mlps.{i}.", mask_tok[:, i], 1) for i in range(4)], 1)
    masks = (hyper @ x.flatten(2)).reshape(n, 4, x.shape[2], x.shape[3])
    iou = _ffn(sd, "mask_decoder.iou_prediction_head.", iou_tok, 1)
    if return_all:
        return masks, iou
    return masks[:, 0], iou[:, 0]


def postprocess(lowres, resized_hw, orig_hw, target=1024):
    """Sam.postprocess_masks + threshold: lowres f32 [n,256,256] -> bool [n,H,W]."""
    m = F.interpolate(lowres[:, None], (target, target), mode="bilinear", align_corners=False)
    m = m[..., :resized_hw[0], :resized_hw[1]]
    m = F.interpolate(m, orig_hw, mode="bilinear", align_corners=False)
    return m[:, 0] > 0.0


def scale_box(box_xyxy, orig_hw, resized_hw):
    """ResizeLongestSide.apply_boxes: coords * (new/old) per axis (float64 in segment_anything, then float32 tensor)."""
    b = np.asarray(box_xyxy, np.float64).reshape(-1, 2, 2).copy()
    b[..., 0] = b[..., 0] * (resized_hw[1] / orig_hw[1])
    b[..., 1] = b[..., 1] * (resized_hw[0] / orig_hw[0])
    return b.reshape(-1, 4).astype(np.float32)
