"""Pins oracle.sam_vit against transformers' SamModel.vision_encoder (architecture-equivalent to segment_anything's
ImageEncoderViT, which the sam3 service would load) on a small config that has both windowed (padded) and global layers."""
import numpy as np
import pytest
import torch

from lmx import sam, weights
from oracle import sam_vit as OV

transformers = pytest.importorskip("transformers")


def test_sam_vit_oracle_matches_transformers():
    from transformers import SamVisionConfig, SamVisionModel

    cfg = sam.SamVitConfig(hidden=128, layers=3, heads=2, mlp=256, global_idx=(1,), window=14, image=512)  # grid 32 -> pad to 42
    sd = weights.synth_state_dict(sam.vit_param_spec(cfg), seed=61)
    vc = SamVisionConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, mlp_dim=cfg.mlp,
                         global_attn_indexes=list(cfg.global_idx), window_size=cfg.window, image_size=cfg.image, patch_size=16,
                         output_channels=cfg.out_ch)
    vc._attn_implementation = "eager"
    m = SamVisionModel(vc).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 3, cfg.image, cfg.image)).astype(np.float32))
    with torch.no_grad():
        ref = m(pixel_values=x).last_hidden_state
        got = OV.encoder_forward(cfg, sd, x)
    assert got.shape == ref.shape == (2, 256, 32, 32)
    assert torch.allclose(got, ref, atol=1e-4, rtol=1e-4), float((got - ref).abs().max())
