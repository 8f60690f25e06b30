"""SAM prompt encoder + mask decoder + mask post-processing on liblmx — the ``predictor.predict(box=..., multimask_output=False)``
half of services/sam3-pipeline/app/main.py:83-89.  The decoder is SAM v1's (the code
path the service calls); it consumes a [n,64,64,256] image embedding, so it serves the Hiera FPN level-2 output
(BASELINE cfg#3/#5) and a SAM ViT neck alike.

Launch sequence per batch of n frames (all through the C-ABI; activations f32 in HBM, GEMM operands f16):
  prompt : 2 box corners -> random-Fourier PE (host, 2x256 floats per frame) + corner embeddings
  decoder: two-way transformer on 7 tokens x 4096 image tokens (GEMMs batched over frames; flash attention kernel with
           Tq=7/Tk=4096 and Tq=4096/Tk=7), then the upscaler as two per-pixel GEMMs (ConvTranspose2d k2 s2 == a 1x1 GEMM
           to 4*Cout channels + pixel shuffle, so LayerNorm2d/GELU stay row-wise and the shuffle is folded into the
           final hyper-network dot product), lmx_k_hyper_mask -> 256x256 logits of mask 0
  post   : lmx_k_mask_post: bilinear 256->1024, crop, bilinear -> frame size, >0, plus area / centroid sums / bounding box.
"""
import math

import numpy as np
import torch

from . import kernels as K

D = 256
HEADS = 8


def param_spec():
    """Ordered {transformers SamModel parameter name: (shape, init kind)} for the prompt encoder + mask decoder."""
    s = {}
    s["shared_image_embedding.positional_embedding"] = ((2, 128), "tok")
    s["prompt_encoder.shared_embedding.positional_embedding"] = ((2, 128), "tok")
    s["prompt_encoder.no_mask_embed.weight"] = ((1, D), "tok")
    for i in range(4):
        s[f"prompt_encoder.point_embed.{i}.weight"] = ((1, D), "tok")
    s["prompt_encoder.not_a_point_embed.weight"] = ((1, D), "tok")
    s["mask_decoder.iou_token.weight"] = ((1, D), "tok")
    s["mask_decoder.mask_tokens.weight"] = ((4, D), "tok")

    def attn(p, inner):
        for n_ in ("q_proj", "k_proj", "v_proj"):
            s[p + n_ + ".weight"] = ((inner, D), "w")
            s[p + n_ + ".bias"] = ((inner,), "b")
        s[p + "out_proj.weight"] = ((D, inner), "w")
        s[p + "out_proj.bias"] = ((D,), "b")

    def ln(p, d=D):
        s[p + "weight"] = ((d,), "g")
        s[p + "bias"] = ((d,), "b")

    for i in range(2):
        p = f"mask_decoder.transformer.layers.{i}."
        attn(p + "self_attn.", D)
        ln(p + "layer_norm1.")
        attn(p + "cross_attn_token_to_image.", D // 2)
        ln(p + "layer_norm2.")
        s[p + "mlp.lin1.weight"] = ((2048, D), "w")
        s[p + "mlp.lin1.bias"] = ((2048,), "b")
        s[p + "mlp.lin2.weight"] = ((D, 2048), "w")
        s[p + "mlp.lin2.bias"] = ((D,), "b")
        ln(p + "layer_norm3.")
        ln(p + "layer_norm4.")
        attn(p + "cross_attn_image_to_token.", D // 2)
    attn("mask_decoder.transformer.final_attn_token_to_image.", D // 2)
    ln("mask_decoder.transformer.layer_norm_final_attn.")
    s["mask_decoder.upscale_conv1.weight"] = ((D, 64, 2, 2), "w")
    s["mask_decoder.upscale_conv1.bias"] = ((64,), "b")
    s["mask_decoder.upscale_conv2.weight"] = ((64, 32, 2, 2), "w")
    s["mask_decoder.upscale_conv2.bias"] = ((32,), "b")
    ln("mask_decoder.upscale_layer_norm.", 64)

    def ffn(p, hid, out):
        s[p + "proj_in.weight"] = ((hid, D), "w")
        s[p + "proj_in.bias"] = ((hid,), "b")
        s[p + "proj_out.weight"] = ((out, hid), "w")
        s[p + "proj_out.bias"] = ((out,), "b")
        s[p + "layers.0.weight"] = ((hid, hid), "w")
        s[p + "layers.0.bias"] = ((hid,), "b")

    for i in range(4):
        ffn(f"mask_decoder.output_hypernetworks_mlps.{i}.", D, 32)
    ffn("mask_decoder.iou_prediction_head.", 256, 4)
    return s


def synthetic_state_dict(seed):
    """Synthetic decoder weights.  SAM has ONE random-Fourier matrix (segment_anything's pe_layer) used for both the
    prompt points and the dense image PE; transformers stores it under two tied names, so both get the same values."""
    from . import weights

    sd = weights.synth_state_dict(param_spec(), seed)
    sd["prompt_encoder.shared_embedding.positional_embedding"] = sd["shared_image_embedding.positional_embedding"].copy()
    return sd


class MaskDecoder:
    """Device-resident SAM prompt encoder + mask decoder.  ``predict(emb, boxes, frame_hw, resized_hw)`` -> dict(mask u8
    [n,h,w], stats int64 [n,8], iou f32 [n], lowres f32 [n,256,256])."""

    def __init__(self, state_dict, device="cuda", image_size=1024, grid=64, precision="exact"):
        """precision: the default plan of predict() — "exact" (services, adapters, the reference schedule: f32 activations,
        22-bit GEMM operands, f32 attention; masks within IoU 0.9995 of the fp32 path) or "f16" (the dense throughput
        schedule: f16 GEMM operands)."""
        if precision not in ("exact", "f16"):
            raise ValueError(f"precision {precision!r}: expected 'exact' or 'f16'")
        self.precision = precision
        self.device = torch.device(device)
        self.S, self.G = image_size, grid
        dev = self.device
        sd = state_dict
        self._sd = {k: np.asarray(v, np.float32) for k, v in sd.items() if k.startswith("mask_decoder.")}
        self._wx = {}

        def t32(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        def t16(a):
            return t32(a).to(torch.float16).contiguous()

        gauss = sd["shared_image_embedding.positional_embedding"].astype(np.float32)
        self.gauss = t32(gauss)
        self.corner = t32(np.concatenate([sd["prompt_encoder.point_embed.2.weight"], sd["prompt_encoder.point_embed.3.weight"]], 0))
        # dense image PE (get_image_wide_positional_embeddings), a constant of the weights: f32 torch ops on the host once
        g = torch.ones((grid, grid), dtype=torch.float32)
        yx = torch.stack([(g.cumsum(1) - 0.5) / grid, (g.cumsum(0) - 0.5) / grid], -1)
        c = 2 * np.pi * ((2 * yx - 1) @ torch.from_numpy(gauss))
        self.key_pe = torch.cat([torch.sin(c), torch.cos(c)], -1).reshape(grid * grid, D).contiguous().to(dev)
        self.no_mask = t32(sd["prompt_encoder.no_mask_embed.weight"])                          # [1,256]
        self.out_tokens = t32(np.concatenate([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], 0))

        def attn(p):
            return {n_: (t16(sd[p + n_ + ".weight"]), t32(sd[p + n_ + ".bias"])) for n_ in ("q_proj", "k_proj", "v_proj", "out_proj")}

        def ln(p):
            return t32(sd[p + "weight"]), t32(sd[p + "bias"])

        self.layers = []
        for i in range(2):
            p = f"mask_decoder.transformer.layers.{i}."
            self.layers.append(dict(sa=attn(p + "self_attn."), ln1=ln(p + "layer_norm1."),
                                    t2i=attn(p + "cross_attn_token_to_image."), ln2=ln(p + "layer_norm2."),
                                    w1=(t16(sd[p + "mlp.lin1.weight"]), t32(sd[p + "mlp.lin1.bias"])),
                                    w2=(t16(sd[p + "mlp.lin2.weight"]), t32(sd[p + "mlp.lin2.bias"])),
                                    ln3=ln(p + "layer_norm3."), ln4=ln(p + "layer_norm4."),
                                    i2t=attn(p + "cross_attn_image_to_token.")))
        self.final = attn("mask_decoder.transformer.final_attn_token_to_image.")
        self.ln_final = ln("mask_decoder.transformer.layer_norm_final_attn.")
        # ConvTranspose2d(k2,s2) as a per-pixel GEMM: Wg[(dy,dx,co), ci] = W[ci,co,dy,dx]; bias repeated per quadrant
        w1 = sd["mask_decoder.upscale_conv1.weight"]
        w2 = sd["mask_decoder.upscale_conv2.weight"]
        self.up1 = (t16(np.transpose(w1, (2, 3, 1, 0)).reshape(4 * w1.shape[1], w1.shape[0])), t32(np.tile(sd["mask_decoder.upscale_conv1.bias"], 4)))
        self.up2 = (t16(np.transpose(w2, (2, 3, 1, 0)).reshape(4 * w2.shape[1], w2.shape[0])), t32(np.tile(sd["mask_decoder.upscale_conv2.bias"], 4)))
        self.up_ln = ln("mask_decoder.upscale_layer_norm.")

        def ffn(p):
            return [(t16(sd[p + n_ + ".weight"]), t32(sd[p + n_ + ".bias"])) for n_ in ("proj_in", "layers.0", "proj_out")]

        self.hyper0 = ffn("mask_decoder.output_hypernetworks_mlps.0.")
        self.iou_head = ffn("mask_decoder.iou_prediction_head.")

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _attn(self, W, q16, k16, v16, n, tq, tk, res):
        """SamAttention: q/k/v projections (f16), flash attention over 8 heads, out projection (+res) in f32."""
        q = K.gemm(q16, *W["q_proj"])
        k = K.gemm(k16, *W["k_proj"])
        v = K.gemm(v16, *W["v_proj"])
        inner = q.shape[1]
        hd = inner // HEADS
        a = torch.empty((n * tq, inner), dtype=torch.float16, device=q.device)
        K.attention(q, k, v, a, n, HEADS, tq, tk, hd, hd ** -0.5)
        return K.gemm(a, W["out_proj"][0], bias=W["out_proj"][1], res=res, out_dtype=torch.float32)

    @staticmethod
    def _ffn(layers, x16):
        h = K.gemm(x16, *layers[0], act=K.ACT_RELU)
        h = K.gemm(h, *layers[1], act=K.ACT_RELU)
        return K.gemm(h, *layers[2], out_dtype=torch.float32)

    # ---- forward ---------------------------------------------------------------------------------------------
    def lowres(self, emb, sparse):
        """emb f16/f32 [n*G*G, 256] (NHWC rows), sparse f32 [n,2,256] -> (logits f32 [n,4G,4G] of mask 0, iou f32 [n])."""
        n = sparse.shape[0]
        T, P = 7, self.G * self.G
        eps = 1e-6
        tokens = torch.cat([self.out_tokens[None].expand(n, -1, -1), sparse], dim=1).reshape(n * T, D).contiguous()
        keys = K.add_bcast(emb, self.no_mask)                 # image embedding + dense no-mask embedding, f32
        queries, qpe = tokens, tokens
        for i, L in enumerate(self.layers):
            if i == 0:
                q16 = K.cast_f16(queries)
                queries = self._attn(L["sa"], q16, q16, q16, n, T, T, None)
            else:
                q16 = K.add_bcast(queries, qpe, out_dtype=torch.float16)
                queries = self._attn(L["sa"], q16, q16, K.cast_f16(queries), n, T, T, queries)
            queries = K.layernorm(queries, *L["ln1"], eps, out_dtype=torch.float32)
            q16 = K.add_bcast(queries, qpe, out_dtype=torch.float16)
            k16 = K.add_bcast(keys, self.key_pe, out_dtype=torch.float16)
            queries = self._attn(L["t2i"], q16, k16, K.cast_f16(keys), n, T, P, queries)
            queries = K.layernorm(queries, *L["ln2"], eps, out_dtype=torch.float32)
            h = K.gemm(K.cast_f16(queries), *L["w1"], act=K.ACT_RELU)
            queries = K.gemm(h, *L["w2"], res=queries, out_dtype=torch.float32)
            queries = K.layernorm(queries, *L["ln3"], eps, out_dtype=torch.float32)
            q16 = K.add_bcast(queries, qpe, out_dtype=torch.float16)
            # (keys have not changed since the token-to-image attention above: k16 is still keys + key_pe)
            keys = self._attn(L["i2t"], k16, q16, K.cast_f16(queries), n, P, T, keys)
            keys = K.layernorm(keys, *L["ln4"], eps, out_dtype=torch.float32)
        q16 = K.add_bcast(queries, qpe, out_dtype=torch.float16)
        k16 = K.add_bcast(keys, self.key_pe, out_dtype=torch.float16)
        keys16 = K.cast_f16(keys)  # the final attention's values and the upscaler's input
        queries = self._attn(self.final, q16, k16, keys16, n, T, P, queries)
        queries = K.layernorm(queries, *self.ln_final, 1e-5, out_dtype=torch.float32)
        q3 = queries.view(n, T, D)
        iou_tok = K.cast_f16(q3[:, 0].contiguous())
        mask_tok = K.cast_f16(q3[:, 1].contiguous())
        # upscaler: per-pixel GEMMs, LayerNorm2d+GELU row-wise on the [.., quadrant, 64] view, pixel shuffle deferred
        u = K.gemm(keys16, *self.up1, out_dtype=torch.float32)                       # [n*P, 4*64]
        u = K.layernorm(u.view(n * P * 4, 64), *self.up_ln, eps, act=K.ACT_GELU)      # f16 [n*P*4, 64]
        u = K.gemm(u, *self.up2, act=K.ACT_GELU)                                     # f16 [n*P*4, 4*32]
        hyper = self._ffn(self.hyper0, mask_tok)                                     # f32 [n,32]
        logits = K.hyper_mask(u, hyper, n, self.G, 32)
        iou = self._ffn(self.iou_head, iou_tok)[:, 0]
        return logits, iou

    # ---- exact plan: f32 activations, x3 operands (csrc/exact.hip), f32 attention ---------------------------------------
    def _lin(self, x, name, act_in=K.ACT_NONE, act=K.ACT_NONE, res=None, w2d=None, bias=None):
        """y = act(act_in(x) @ W^T + b) (+ res) with 22-bit operands: x f32 [rows, K] is split into x3 rows (after act_in),
        W into [whi | whi/2048 | wlo] rows pre-scaled by powers of two (lmx.exact.split_rows_x3), ONE lmx_k_gemm launch over
        3K with f32 output.  act may be NONE or RELU (it commutes with the positive row scale)."""
        from .exact import split_rows_x3

        if name not in self._wx:
            w = self._sd[name + ".weight"] if w2d is None else w2d
            b = self._sd[name + ".bias"] if bias is None else bias
            x3, sc, e = split_rows_x3(w, [w.shape[1]])
            dev = self.device
            self._wx[name] = (torch.from_numpy(x3).to(dev), torch.from_numpy(np.ldexp(b.astype(np.float32), e).astype(np.float32)).to(dev),
                              torch.from_numpy(sc).to(dev))
        w3, b3, sc = self._wx[name]
        return K.gemm(K.split3_rows(x, act_in), w3, bias=b3, act=act, scale=sc, res=res, out_dtype=torch.float32)

    def _attn_exact(self, p, q_in, k_in, v_in, n, tq, tk, res):
        q = self._lin(q_in, p + "q_proj")
        k = self._lin(k_in, p + "k_proj")
        v = self._lin(v_in, p + "v_proj")
        hd = q.shape[1] // HEADS
        a = K.attention_f32(q, k, v, n, HEADS, tq, tk, hd, hd ** -0.5)
        return self._lin(a, p + "out_proj", res=res)

    def _ffn_exact(self, p, x):
        h = self._lin(x, p + "proj_in", act=K.ACT_RELU)
        h = self._lin(h, p + "layers.0", act=K.ACT_RELU)
        return self._lin(h, p + "proj_out")

    def lowres_exact(self, emb, sparse):
        """The exact plan of lowres(): same launch structure, every tensor f32 (TF:models/sam/modeling_sam.py:432-543)."""
        n = sparse.shape[0]
        T, P = 7, self.G * self.G
        eps = 1e-6
        f32 = torch.float32
        tokens = torch.cat([self.out_tokens[None].expand(n, -1, -1), sparse], dim=1).reshape(n * T, D).contiguous()
        keys = K.add_bcast(emb, self.no_mask)                 # image embedding + dense no-mask embedding, f32
        queries, qpe = tokens, tokens
        for i in range(2):
            p = f"mask_decoder.transformer.layers.{i}."
            L = self.layers[i]
            if i == 0:
                queries = self._attn_exact(p + "self_attn.", queries, queries, queries, n, T, T, None)
            else:
                q_ = K.add_bcast(queries, qpe, out_dtype=f32)
                queries = self._attn_exact(p + "self_attn.", q_, q_, queries, n, T, T, queries)
            queries = K.layernorm(queries, *L["ln1"], eps, out_dtype=f32)
            q_ = K.add_bcast(queries, qpe, out_dtype=f32)
            k_ = K.add_bcast(keys, self.key_pe, out_dtype=f32)
            queries = self._attn_exact(p + "cross_attn_token_to_image.", q_, k_, keys, n, T, P, queries)
            queries = K.layernorm(queries, *L["ln2"], eps, out_dtype=f32)
            h = self._lin(queries, p + "mlp.lin1", act=K.ACT_RELU)
            queries = self._lin(h, p + "mlp.lin2", res=queries)
            queries = K.layernorm(queries, *L["ln3"], eps, out_dtype=f32)
            q_ = K.add_bcast(queries, qpe, out_dtype=f32)
            keys = self._attn_exact(p + "cross_attn_image_to_token.", k_, q_, queries, n, P, T, keys)
            keys = K.layernorm(keys, *L["ln4"], eps, out_dtype=f32)
        p = "mask_decoder.transformer."
        q_ = K.add_bcast(queries, qpe, out_dtype=f32)
        k_ = K.add_bcast(keys, self.key_pe, out_dtype=f32)
        queries = self._attn_exact(p + "final_attn_token_to_image.", q_, k_, keys, n, T, P, queries)
        queries = K.layernorm(queries, *self.ln_final, 1e-5, out_dtype=f32)
        q3 = queries.view(n, T, D)
        iou_tok, mask_tok = q3[:, 0].contiguous(), q3[:, 1].contiguous()
        # upscaler: ConvTranspose2d(k2, s2) as per-pixel GEMMs (weights re-laid as in __init__), LayerNorm2d row-wise on the
        # [.., quadrant, 64] view, exact GELUs folded into the next split / the final dot product
        w1 = self._sd["mask_decoder.upscale_conv1.weight"]
        w2 = self._sd["mask_decoder.upscale_conv2.weight"]
        u = self._lin(keys, "mask_decoder.upscale_conv1", w2d=np.transpose(w1, (2, 3, 1, 0)).reshape(4 * w1.shape[1], w1.shape[0]),
                      bias=np.tile(self._sd["mask_decoder.upscale_conv1.bias"], 4))                    # [n*P, 4*64]
        u = K.layernorm(u.view(n * P * 4, 64), *self.up_ln, eps, out_dtype=f32)
        u = self._lin(u, "mask_decoder.upscale_conv2", act_in=K.ACT_GELU,
                      w2d=np.transpose(w2, (2, 3, 1, 0)).reshape(4 * w2.shape[1], w2.shape[0]),
                      bias=np.tile(self._sd["mask_decoder.upscale_conv2.bias"], 4))                    # [n*P*4, 4*32], pre-GELU
        hyper = self._ffn_exact("mask_decoder.output_hypernetworks_mlps.0.", mask_tok)                # f32 [n,32]
        logits = K.hyper_mask_f32(u, hyper, n, self.G, 32, act=K.ACT_GELU)
        iou = self._ffn_exact("mask_decoder.iou_prediction_head.", iou_tok)[:, 0]
        return logits, iou

    def predict(self, emb, boxes, frame_hw, resized_hw, precision=None):
        """emb [n*G*G,256] rows of the NHWC image embedding; boxes f32 [n,>=4] device, xyxy in FRAME pixels.
        precision: None = this decoder's default plan, "exact" or "f16" (constructor docstring)."""
        h, w = frame_hw
        nh, nw = resized_hw
        sparse = K.prompt_box(boxes, nw / w, nh / h, float(self.S), self.gauss, self.corner)
        precision = precision or self.precision
        if precision not in ("exact", "f16"):
            raise ValueError(f"precision {precision!r}: expected 'exact' or 'f16'")
        logits, iou = (self.lowres_exact if precision == "exact" else self.lowres)(emb, sparse)
        mask, stats = K.mask_post(logits, self.S, nh, nw, h, w)
        return dict(mask=mask, stats=stats, iou=iou, lowres=logits)
