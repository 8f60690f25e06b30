"""Pins oracle.vit against the transformers model classes installed in the container (the importable
oracle for the DINO path), on small configs built from config objects with the build's own synthetic weights."""
import numpy as np
import pytest
import torch

from lmx import dino, weights
from oracle import vit

transformers = pytest.importorskip("transformers")


def _hf_v3(cfg, sd):
    from transformers import DINOv3ViTConfig, DINOv3ViTModel

    c = DINOv3ViTConfig(hidden_size=cfg.hidden, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers,
                        num_attention_heads=cfg.heads, num_register_tokens=cfg.registers, patch_size=cfg.patch,
                        layer_norm_eps=cfg.eps, rope_theta=cfg.rope_theta, image_size=cfg.image,
                        attn_implementation="eager")
    m = DINOv3ViTModel(c).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m


def _hf_v2(cfg, sd):
    from transformers import Dinov2Config, Dinov2Model

    c = Dinov2Config(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                     mlp_ratio=cfg.mlp // cfg.hidden, patch_size=cfg.patch, image_size=cfg.pos_grid * cfg.patch,
                     layer_norm_eps=cfg.eps, attn_implementation="eager")
    m = Dinov2Model(c).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m


@pytest.mark.parametrize("size", [(224, 224), (96, 160)])
def test_dinov3_oracle_matches_transformers(size):
    cfg = dino.DinoConfig(hidden=128, layers=3, heads=2, mlp=256, registers=4)
    sd = weights.synth_state_dict(dino.param_spec(cfg), seed=11)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 3) + size).astype(np.float32))
    with torch.no_grad():
        ref = _hf_v3(cfg, sd)(pixel_values=x).last_hidden_state
        got = vit.dinov3_forward(cfg, sd, x)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), float((got - ref).abs().max())


@pytest.mark.parametrize("pos_grid", [16, 37])
def test_dinov2_oracle_matches_transformers(pos_grid):
    cfg = dino.DinoConfig(arch="dinov2", hidden=96, layers=2, heads=3, mlp=384, patch=14, registers=0, eps=1e-6,
                          pos_grid=pos_grid)
    sd = weights.synth_state_dict(dino.param_spec(cfg), seed=12)
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 3, 224, 224)).astype(np.float32))
    with torch.no_grad():
        ref = _hf_v2(cfg, sd)(pixel_values=x).last_hidden_state
        got = vit.dinov2_forward(cfg, sd, x)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), float((got - ref).abs().max())


def test_host_tables_match_oracle():
    cfg = dino.DinoConfig(hidden=128, layers=1, heads=2, mlp=256)
    c1, s1 = dino.rope_tables(cfg, 14, 14)
    c2, s2 = vit.rope_tables(cfg.head_dim, cfg.rope_theta, 14, 14)
    assert torch.equal(c1, c2) and torch.equal(s1, s2)
