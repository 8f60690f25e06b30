"""Pins oracle.sam_decoder against transformers' SamModel prompt encoder + mask decoder (the importable, architecture-
equivalent implementation of segment_anything's) on identical image embeddings and box prompts."""
import numpy as np
import pytest
import torch

from lmx import sam_decoder, weights
from oracle import sam_decoder as OD

transformers = pytest.importorskip("transformers")


def _hf(sd):
    from transformers import SamConfig, SamModel

    c = SamConfig()
    c.vision_config.num_hidden_layers = 1
    c.vision_config.hidden_size = 64
    c.vision_config.num_attention_heads = 2
    c.vision_config.mlp_dim = 128
    c.vision_config.global_attn_indexes = [0]
    c.mask_decoder_config._attn_implementation = "eager"
    m = SamModel(c).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith("vision_encoder.") or "mask_embed" in k for k in missing), (missing, unexpected)
    return m


def test_prompt_encoder_and_mask_decoder_match_transformers():
    sd = sam_decoder.synthetic_state_dict(41)
    m = _hf(sd)
    rng = np.random.default_rng(0)
    emb = torch.from_numpy(rng.standard_normal((2, 256, 64, 64)).astype(np.float32))
    boxes = torch.tensor([[100.0, 150.0, 700.0, 500.0], [12.5, 30.0, 1000.0, 560.0]])
    with torch.no_grad():
        sparse_ref, dense_ref = m.prompt_encoder(input_points=None, input_labels=None, input_boxes=boxes[:, None], input_masks=None)
        pe_ref = m.get_image_wide_positional_embeddings()
        masks_ref, iou_ref = m.mask_decoder(image_embeddings=emb, image_positional_embeddings=pe_ref.repeat(2, 1, 1, 1),
                                            sparse_prompt_embeddings=sparse_ref, dense_prompt_embeddings=dense_ref,
                                            multimask_output=False)
        sparse = OD.prompt_encode_box(sd, boxes)
        low, iou = OD.mask_decode(sd, emb, sparse)
    assert torch.allclose(sparse, sparse_ref[:, 0], atol=1e-5)
    assert torch.allclose(OD.image_pe(sd).permute(2, 0, 1), pe_ref[0], atol=1e-5)
    assert torch.allclose(low, masks_ref[:, 0, 0], atol=2e-4, rtol=1e-4), float((low - masks_ref[:, 0, 0]).abs().max())
    assert torch.allclose(iou, iou_ref[:, 0, 0], atol=1e-4)


def test_postprocess_shapes_and_box_scaling():
    low = torch.from_numpy(np.random.default_rng(1).standard_normal((1, 256, 256)).astype(np.float32))
    m = OD.postprocess(low, (576, 1024), (1080, 1920))
    assert m.shape == (1, 1080, 1920) and m.dtype == torch.bool
    b = OD.scale_box([300, 150, 1200, 900], (1080, 1920), (576, 1024))
    assert np.allclose(b, [[160.0, 80.0, 640.0, 480.0]])
