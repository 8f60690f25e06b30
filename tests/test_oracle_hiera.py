"""Pins oracle.hiera against transformers' Sam2VisionModel (the importable implementation of the SAM2 Hiera trunk +
FPN neck) with the build's synthetic weights; also checks lmx.sam's host-side plan against it."""
import numpy as np
import pytest
import torch

from lmx import sam, weights
from oracle import hiera as OH

transformers = pytest.importorskip("transformers")


def _hf(cfg, sd):
    from transformers import Sam2HieraDetConfig, Sam2VisionConfig, Sam2VisionModel

    hc = Sam2HieraDetConfig(hidden_size=cfg.hidden, num_attention_heads=cfg.heads[0], blocks_per_stage=list(cfg.blocks),
                            embed_dim_per_stage=list(cfg.dims), num_attention_heads_per_stage=list(cfg.heads),
                            window_size_per_stage=list(cfg.windows), global_attention_blocks=list(cfg.global_blocks),
                            window_positional_embedding_background_size=list(cfg.pos_bkg), image_size=[cfg.image, cfg.image],
                            layer_norm_eps=cfg.eps)
    vc = Sam2VisionConfig(backbone_config=hc, backbone_channel_list=list(reversed(cfg.dims)), fpn_hidden_size=cfg.fpn_dim,
                          fpn_top_down_levels=list(cfg.fpn_top_down))
    vc._attn_implementation = "eager"
    hc._attn_implementation = "eager"
    m = Sam2VisionModel(vc).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m


def _tiny(image=256):
    return sam.HieraConfig(hidden=16, blocks=(1, 2, 3, 2), dims=(16, 32, 64, 128), heads=(1, 2, 4, 8), windows=(8, 4, 14, 7),
                           global_blocks=(4,), pos_bkg=(7, 7), fpn_dim=32, image=image)


@pytest.mark.parametrize("image", [256, 320])
def test_hiera_oracle_matches_transformers(image):
    cfg = _tiny(image)  # 256: grids 64/32/16/8 -> window 14 pads 16->28, window 7 pads 8->14; 320: 80/40/20/10
    sd = weights.synth_state_dict(sam.param_spec(cfg), seed=31)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 3, image, image)).astype(np.float32))
    with torch.no_grad():
        ref = _hf(cfg, sd)(pixel_values=x)
        fpn, outs = OH.encoder_forward(cfg, sd, x)
    assert torch.allclose(outs[-1], ref.last_hidden_state, atol=5e-5, rtol=1e-4), float((outs[-1] - ref.last_hidden_state).abs().max())
    assert len(fpn) == len(ref.fpn_hidden_states) == 3
    for a, b in zip(fpn, ref.fpn_hidden_states):
        assert a.shape == b.shape and torch.allclose(a, b, atol=5e-5, rtol=1e-4), float((a - b).abs().max())
    pe = ref.fpn_position_encoding[2][0].permute(1, 2, 0)
    assert torch.allclose(OH.sine_pos(cfg.fpn_dim, pe.shape[0], pe.shape[1]), pe, atol=1e-6)


def test_block_plan_is_hiera_b_plus():
    plan = sam.hiera_b_plus().block_plan()
    assert len(plan) == 24
    assert plan[0] == (112, 112, 2, 8, 0) and plan[2] == (112, 224, 4, 8, 2) and plan[5] == (224, 448, 8, 4, 2)
    assert plan[12] == (448, 448, 8, 0, 0) and plan[21] == (448, 896, 16, 14, 2) and plan[23] == (896, 896, 16, 7, 0)
    assert OH.hiera_b_plus().block_plan() == plan
    n_params = sum(int(np.prod(s)) for s, _ in sam.param_spec(sam.hiera_b_plus()).values())
    assert abs(n_params - 69.11e6) < 0.05e6  # 69.11 M parameters for Hiera-B+ trunk + neck


def test_resize_longest_side():
    assert sam.resize_longest_side(1080, 1920) == (576, 1024)
    assert sam.resize_longest_side(720, 1280) == (576, 1024)
    assert sam.resize_longest_side(1000, 600) == (1024, 614)
