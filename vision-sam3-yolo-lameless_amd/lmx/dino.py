"""DINO ViT embedder on liblmx — the model call of services/dinov3-pipeline/app/main.py:95-115
(``processor(images=...)`` -> ``model(**inputs).last_hidden_state.mean(dim=1)``).

Two architectures behind one launch sequence:
  * ``dinov2``  — what the service loads by default (facebook/dinov2-base): patch 14, learned position table
                   (bicubic-interpolated to the input grid once, at load), separate q/k/v with bias, eps 1e-6.
  * ``dinov3``  — the BASELINE config (ViT-L/16): patch 16, CLS + 4 register tokens, no position table, RoPE on the
                   patch tokens of q/k, key without bias, eps 1e-5.
Layer = LN -> fused qkv GEMM -> (RoPE) -> flash attention -> out-proj GEMM with LayerScale+residual epilogue ->
LN -> fc1 GEMM with GELU epilogue -> fc2 GEMM with LayerScale+residual epilogue.  The residual stream is f32 in
HBM; every GEMM/attention operand is f16 with f32 accumulation.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import kernels as K
from . import resample


@dataclass
class DinoConfig:
    arch: str = "dinov3"          # "dinov3" | "dinov2"
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    mlp: int = 4096
    patch: int = 16
    registers: int = 4
    eps: float = 1e-5
    rope_theta: float = 100.0
    image: int = 224              # crop size fed to the network
    resize_edge: int = 256        # shortest-edge resize before the crop
    pos_grid: int = 37            # dinov2 only: side of the learned position grid (518/14)

    @property
    def head_dim(self):
        return self.hidden // self.heads

    @property
    def n_prefix(self):
        return 1 + (self.registers if self.arch == "dinov3" else 0)

    @property
    def grid(self):
        return self.image // self.patch

    @property
    def tokens(self):
        return self.n_prefix + self.grid * self.grid


def dinov3_vitl16():
    return DinoConfig()


def dinov2_base():
    return DinoConfig(arch="dinov2", hidden=768, layers=12, heads=12, mlp=3072, patch=14, registers=0, eps=1e-6)


def param_spec(cfg):
    """Ordered {HF parameter name: (shape, init kind)} — the names transformers' DINOv3ViTModel / Dinov2Model use."""
    D, I, P = cfg.hidden, cfg.mlp, cfg.patch
    s = {}
    if cfg.arch == "dinov3":
        s["embeddings.cls_token"] = ((1, 1, D), "tok")
        s["embeddings.mask_token"] = ((1, 1, D), "zero")
        s["embeddings.register_tokens"] = ((1, cfg.registers, D), "tok")
        s["embeddings.patch_embeddings.weight"] = ((D, 3, P, P), "w")
        s["embeddings.patch_embeddings.bias"] = ((D,), "b")
        for i in range(cfg.layers):
            p = f"model.layer.{i}."
            s[p + "norm1.weight"] = ((D,), "g")
            s[p + "norm1.bias"] = ((D,), "b")
            s[p + "attention.k_proj.weight"] = ((D, D), "w")
            s[p + "attention.v_proj.weight"] = ((D, D), "w")
            s[p + "attention.v_proj.bias"] = ((D,), "b")
            s[p + "attention.q_proj.weight"] = ((D, D), "w")
            s[p + "attention.q_proj.bias"] = ((D,), "b")
            s[p + "attention.o_proj.weight"] = ((D, D), "w")
            s[p + "attention.o_proj.bias"] = ((D,), "b")
            s[p + "layer_scale1.lambda1"] = ((D,), "ls")
            s[p + "norm2.weight"] = ((D,), "g")
            s[p + "norm2.bias"] = ((D,), "b")
            s[p + "mlp.up_proj.weight"] = ((I, D), "w")
            s[p + "mlp.up_proj.bias"] = ((I,), "b")
            s[p + "mlp.down_proj.weight"] = ((D, I), "w")
            s[p + "mlp.down_proj.bias"] = ((D,), "b")
            s[p + "layer_scale2.lambda1"] = ((D,), "ls")
        s["norm.weight"] = ((D,), "g")
        s["norm.bias"] = ((D,), "b")
    else:
        s["embeddings.cls_token"] = ((1, 1, D), "tok")
        s["embeddings.mask_token"] = ((1, D), "zero")
        s["embeddings.position_embeddings"] = ((1, 1 + cfg.pos_grid * cfg.pos_grid, D), "tok")
        s["embeddings.patch_embeddings.projection.weight"] = ((D, 3, P, P), "w")
        s["embeddings.patch_embeddings.projection.bias"] = ((D,), "b")
        for i in range(cfg.layers):
            p = f"encoder.layer.{i}."
            s[p + "norm1.weight"] = ((D,), "g")
            s[p + "norm1.bias"] = ((D,), "b")
            for n in ("query", "key", "value"):
                s[p + f"attention.attention.{n}.weight"] = ((D, D), "w")
                s[p + f"attention.attention.{n}.bias"] = ((D,), "b")
            s[p + "attention.output.dense.weight"] = ((D, D), "w")
            s[p + "attention.output.dense.bias"] = ((D,), "b")
            s[p + "layer_scale1.lambda1"] = ((D,), "ls")
            s[p + "norm2.weight"] = ((D,), "g")
            s[p + "norm2.bias"] = ((D,), "b")
            s[p + "mlp.fc1.weight"] = ((I, D), "w")
            s[p + "mlp.fc1.bias"] = ((I,), "b")
            s[p + "mlp.fc2.weight"] = ((D, I), "w")
            s[p + "mlp.fc2.bias"] = ((D,), "b")
            s[p + "layer_scale2.lambda1"] = ((D,), "ls")
        s["layernorm.weight"] = ((D,), "g")
        s["layernorm.bias"] = ((D,), "b")
    return s


def rope_tables(cfg, gh, gw):
    """cos/sin f32 [gh*gw, head_dim] exactly as DINOv3ViTRopePositionEmbedding.forward builds them
    (TF:models/dinov3_vit/modeling_dinov3_vit.py:153-200): f32 torch ops on the host, once per grid shape."""
    hd = cfg.head_dim
    inv_freq = 1 / cfg.rope_theta ** torch.arange(0, 1, 4 / hd, dtype=torch.float32)
    ch = torch.arange(0.5, gh, dtype=torch.float32) / gh
    cw = torch.arange(0.5, gw, dtype=torch.float32) / gw
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    angles = 2 * math.pi * coords[:, :, None] * inv_freq[None, None, :]
    angles = angles.flatten(1, 2).tile(2)
    return torch.cos(angles).contiguous(), torch.sin(angles).contiguous()


def interpolate_pos_embed(pos, grid_in, grid_out):
    """Dinov2Embeddings.interpolate_pos_encoding (TF:models/dinov2/modeling_dinov2.py:57-95): bicubic,
    align_corners=False, computed in f32 on the host once at load.  pos: torch [1, 1+gi*gi, D]."""
    if grid_in == grid_out:
        return pos
    cls, patch = pos[:, :1], pos[:, 1:]
    D = pos.shape[-1]
    patch = patch.reshape(1, grid_in, grid_in, D).permute(0, 3, 1, 2)
    patch = torch.nn.functional.interpolate(patch.to(torch.float32), size=(grid_out, grid_out), mode="bicubic",
                                            align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, -1, D)
    return torch.cat((cls, patch), dim=1)


class DinoEmbedder:
    """Device-resident DINO ViT.  ``embed_patches`` is the network proper (cfg#4 input: already-preprocessed
    frames as a patch matrix), ``embed_frames`` adds the service's preprocessing from raw BGR u8 frames."""

    def __init__(self, cfg, state_dict, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        dev = self.device
        D, P = cfg.hidden, cfg.patch

        def t32(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        def t16(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev).to(torch.float16).contiguous()

        sd = state_dict
        v3 = cfg.arch == "dinov3"
        pe_w = sd["embeddings.patch_embeddings.weight" if v3 else "embeddings.patch_embeddings.projection.weight"]
        pe_b = sd["embeddings.patch_embeddings.bias" if v3 else "embeddings.patch_embeddings.projection.bias"]
        # conv weight [D,3,P,P] -> GEMM weight [D, (ky,kx,c)], K padded to a multiple of 8 with zero columns
        k_raw = P * P * 3
        self.k_pad = (k_raw + 7) // 8 * 8
        w = np.transpose(pe_w, (0, 2, 3, 1)).reshape(D, k_raw)
        if self.k_pad != k_raw:
            w = np.concatenate([w, np.zeros((D, self.k_pad - k_raw), np.float32)], axis=1)
        self.pe_w, self.pe_b = t16(w), t32(pe_b)
        if v3:
            prefix = np.concatenate([sd["embeddings.cls_token"][0], sd["embeddings.register_tokens"][0]], axis=0)
            self.pos = None
            self.rope = tuple(t.to(dev) for t in rope_tables(cfg, cfg.grid, cfg.grid))
        else:
            prefix = sd["embeddings.cls_token"][0]
            pos = interpolate_pos_embed(torch.from_numpy(sd["embeddings.position_embeddings"]), cfg.pos_grid, cfg.grid)
            self.pos = pos[0].to(torch.float32).contiguous().to(dev)
            self.rope = None
        self.prefix = t32(prefix)
        self.layers = []
        for i in range(cfg.layers):
            if v3:
                p = f"model.layer.{i}."
                qw, kw, vw = (sd[p + f"attention.{n}_proj.weight"] for n in "qkv")
                qb, vb = sd[p + "attention.q_proj.bias"], sd[p + "attention.v_proj.bias"]
                kb = sd.get(p + "attention.k_proj.bias", np.zeros(D, np.float32))
                ow, ob = sd[p + "attention.o_proj.weight"], sd[p + "attention.o_proj.bias"]
                w1, b1 = sd[p + "mlp.up_proj.weight"], sd[p + "mlp.up_proj.bias"]
                w2, b2 = sd[p + "mlp.down_proj.weight"], sd[p + "mlp.down_proj.bias"]
            else:
                p = f"encoder.layer.{i}."
                qw, kw, vw = (sd[p + f"attention.attention.{n}.weight"] for n in ("query", "key", "value"))
                qb, kb, vb = (sd[p + f"attention.attention.{n}.bias"] for n in ("query", "key", "value"))
                ow, ob = sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]
                w1, b1 = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]
                w2, b2 = sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]
            self.layers.append(dict(
                g1=t32(sd[p + "norm1.weight"]), b1=t32(sd[p + "norm1.bias"]),
                wqkv=t16(np.concatenate([qw, kw, vw], 0)), bqkv=t32(np.concatenate([qb, kb, vb], 0)),
                wo=t16(ow), bo=t32(ob), ls1=t32(sd[p + "layer_scale1.lambda1"]),
                g2=t32(sd[p + "norm2.weight"]), b2=t32(sd[p + "norm2.bias"]),
                w1=t16(w1), bb1=t32(b1), w2=t16(w2), bb2=t32(b2), ls2=t32(sd[p + "layer_scale2.lambda1"])))
        self.gf = t32(sd["norm.weight" if v3 else "layernorm.weight"])
        self.bf = t32(sd["norm.bias" if v3 else "layernorm.bias"])
        self.lut = t32(resample.norm_lut(resample.IMAGENET_MEAN, resample.IMAGENET_STD))
        self._tabs = {}

    # ---- the network --------------------------------------------------------------------------------------
    def hidden_states(self, patches, B):
        """patches f16 [B*np, k_pad] -> final-LayerNorm'ed tokens f32 [B*T, D]."""
        cfg = self.cfg
        D, H, hd, T = cfg.hidden, cfg.heads, cfg.head_dim, cfg.tokens
        np_ = cfg.grid * cfg.grid
        xp = K.gemm(patches, self.pe_w, bias=self.pe_b)
        x = K.assemble_tokens(xp, self.prefix, self.pos, B, np_, cfg.n_prefix, D)
        scale = hd ** -0.5
        for L in self.layers:
            h = K.layernorm(x, L["g1"], L["b1"], cfg.eps)
            qkv = K.gemm(h, L["wqkv"], bias=L["bqkv"])
            q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            if self.rope is not None:
                K.rope(q, B, T, H, hd, cfg.n_prefix, *self.rope)
                K.rope(k, B, T, H, hd, cfg.n_prefix, *self.rope)
            a = torch.empty((B * T, D), dtype=torch.float16, device=x.device)
            K.attention(q, k, v, a, B, H, T, T, hd, scale)
            K.gemm(a, L["wo"], bias=L["bo"], scale=L["ls1"], res=x, out=x)
            h = K.layernorm(x, L["g2"], L["b2"], cfg.eps)
            u = K.gemm(h, L["w1"], bias=L["bb1"], act=K.ACT_GELU)
            K.gemm(u, L["w2"], bias=L["bb2"], scale=L["ls2"], res=x, out=x)
        return K.layernorm(x, self.gf, self.bf, cfg.eps, out_dtype=torch.float32)

    def embed_patches(self, patches, B):
        """-> f32 [B, D]: mean over ALL tokens of last_hidden_state (dinov3 main.py:113)."""
        y = self.hidden_states(patches, B)
        return K.token_mean(y, B, self.cfg.tokens, self.cfg.hidden)

    # ---- preprocessing (K21) ------------------------------------------------------------------------------
    def _tables(self, h, w):
        key = (h, w)
        if key not in self._tabs:
            nh, nw = resample.shortest_edge_size(h, w, self.cfg.resize_edge)
            dev = self.device

            def up(tab):
                b, k, ks = tab
                return (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks)

            # the horizontal pass always runs: it also does the BGR->RGB swap (identity table if the width is kept)
            th = up(resample.coeff_tables(w, nw, resample.BICUBIC)) if nw != w else up(_identity_table(w))
            tv = up(resample.coeff_tables(h, nh, resample.BICUBIC)) if nh != h else None
            self._tabs[key] = (nh, nw, th, tv)
        return self._tabs[key]

    def preprocess(self, frames_bgr, rgb=False):
        """u8 [B,H,W,3] BGR (cv2 order; `rgb=True`: already RGB, as the PIL image the glue hands the processor) on device
        -> f16 patch matrix [B*np, k_pad]:
        cvtColor(BGR2RGB) -> PIL bicubic shortest-edge resize -> center crop -> /255 -> ImageNet normalise."""
        cfg = self.cfg
        B, h, w, _ = frames_bgr.shape
        nh, nw, th, tv = self._tables(h, w)
        if nh < cfg.image or nw < cfg.image:
            raise K.LmxError(f"frame {h}x{w} resizes to {nh}x{nw}, smaller than the {cfg.image} crop")
        img = K.pil_resize(frames_bgr, nw, nh, th, tv, swap_rb=not rgb)
        top, left = (nh - cfg.image) // 2, (nw - cfg.image) // 2
        return K.patchify_norm(img, top, left, cfg.grid, cfg.grid, cfg.patch, self.lut, k_pad=self.k_pad)

    def embed_frames(self, frames_bgr):
        return self.embed_patches(self.preprocess(frames_bgr), frames_bgr.shape[0])


def _identity_table(n):
    bounds = np.stack([np.arange(n, dtype=np.int32), np.ones(n, np.int32)], 1).reshape(-1)
    kk = np.full((n,), 1 << resample.PRECISION_BITS, dtype=np.int32)
    return bounds, kk, 1
