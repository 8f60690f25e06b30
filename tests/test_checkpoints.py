"""Model selection and checkpoint key maps: the file-name rules of
services/sam3-pipeline/app/main.py:51-72 and services/yolo-pipeline/app/main.py:24-37, and round trips through files
written in the reference's own parameter namings (segment_anything `.pth`, Ultralytics `model.N.*`, a Hugging Face
model directory), read back with loaders that execute nothing from the file.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from lmx import checkpoints as CK
from lmx import dino, sam, sam_decoder, weights, yolo


def _small_sam():
    cfg = sam.SamVitConfig(hidden=64, layers=3, heads=2, mlp=128, global_idx=(1,), window=14, image=512)
    sd = weights.synth_state_dict(sam.vit_param_spec(cfg), 3)
    sd.update(sam_decoder.synthetic_state_dict(4))
    return cfg, sd


def test_sam_model_type_from_file_name():
    assert CK.sam_model_type("sam_vit_h_4b8939.pth") == "vit_h"
    assert CK.sam_model_type("/x/sam_vit_l_0b3195.pth") == "vit_l"
    assert CK.sam_model_type("sam_vit_b_01ec64.pth") == "vit_b"
    assert CK.sam_model_type("my_finetune.pth") == "vit_b"          # anything else: vit_b (main.py:62-63)
    assert CK.sam_model_type("vit_l_then_vit_h.pth") == "vit_h"     # vit_h is tested first (main.py:58)


def test_find_sam_checkpoint(tmp_path):
    assert CK.find_sam_checkpoint(tmp_path / "missing") == (None, None)
    (tmp_path / "notes.txt").write_text("x")
    assert CK.find_sam_checkpoint(tmp_path) == (None, None)            # -> rectangle fallback in the service
    (tmp_path / "sam_vit_l_0b3195.pth").write_bytes(b"")
    p, t = CK.find_sam_checkpoint(tmp_path)
    assert p.name == "sam_vit_l_0b3195.pth" and t == "vit_l"


def test_segment_anything_names_round_trip(tmp_path):
    cfg, sd = _small_sam()
    sa = CK.lmx_to_segment_anything(sd)
    # spot checks against segment_anything's module tree (image_encoder / prompt_encoder / mask_decoder)
    for name in ["image_encoder.pos_embed", "image_encoder.patch_embed.proj.weight", "image_encoder.blocks.0.norm1.weight",
                 "image_encoder.blocks.2.attn.rel_pos_h", "image_encoder.blocks.1.mlp.lin2.bias", "image_encoder.neck.0.weight",
                 "image_encoder.neck.1.bias", "image_encoder.neck.3.weight",
                 "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", "prompt_encoder.point_embeddings.2.weight",
                 "prompt_encoder.no_mask_embed.weight", "mask_decoder.transformer.layers.1.norm4.weight",
                 "mask_decoder.transformer.layers.0.cross_attn_token_to_image.q_proj.weight",
                 "mask_decoder.transformer.norm_final_attn.bias", "mask_decoder.output_upscaling.0.weight",
                 "mask_decoder.output_upscaling.1.weight", "mask_decoder.output_upscaling.3.bias",
                 "mask_decoder.output_hypernetworks_mlps.3.layers.0.weight", "mask_decoder.output_hypernetworks_mlps.0.layers.2.bias",
                 "mask_decoder.iou_prediction_head.layers.1.weight", "mask_decoder.iou_token.weight"]:
        assert name in sa, name
    assert not any(k.startswith(("vision_encoder.", "shared_image_embedding.")) for k in sa)
    back = CK.segment_anything_to_lmx(sa)
    assert set(back) == set(sd)
    for k in sd:
        assert np.array_equal(back[k], sd[k]), k
    # through a file in segment_anything's format: a plain tensor dict saved with torch.save
    path = tmp_path / "sam_vit_b_test.pth"
    torch.save({k: torch.from_numpy(v) for k, v in sa.items()}, path)
    cfg2, sd2 = CK.load_sam_checkpoint(path)
    assert (cfg2.hidden, cfg2.layers, cfg2.heads, cfg2.mlp, tuple(cfg2.global_idx), cfg2.window, cfg2.image) == \
        (cfg.hidden, cfg.layers, cfg.heads, cfg.mlp, tuple(cfg.global_idx), cfg.window, cfg.image)
    for k in sd:
        assert np.array_equal(sd2[k], sd[k]), k
    with pytest.raises(RuntimeError, match="not a vit_h"):
        CK.load_sam_checkpoint(path, "vit_h")   # sam_model_registry["vit_h"](checkpoint=<vit_b file>) fails in the reference too


def test_full_size_sam_configs_are_recognised():
    for mk, name in ((sam.sam_vit_b, "vit_b"), (sam.sam_vit_l, "vit_l"), (sam.sam_vit_h, "vit_h")):
        cfg = mk()
        shapes = {k: np.empty(shape, np.float32) for k, (shape, _) in sam.vit_param_spec(cfg).items() if "layers" not in k
                  or k.endswith(("rel_pos_h", "mlp.lin1.weight"))}
        got = CK.sam_vit_config_from_state_dict(shapes)
        assert (got.hidden, got.layers, got.heads, got.mlp, tuple(got.global_idx)) == \
            (cfg.hidden, cfg.layers, cfg.heads, cfg.mlp, tuple(cfg.global_idx)), name


@pytest.mark.parametrize("scale,nc,kpt", [("n", 80, None), ("s", 3, None), ("n", 1, (17, 3))])
def test_yolo_state_dict_round_trip(tmp_path, scale, nc, kpt):
    from safetensors.numpy import save_file

    cfg = yolo.YoloConfig(scale, nc=nc, kpt_shape=kpt)
    sd = yolo.synthetic_state_dict(cfg, 5)
    assert CK.yolo_config_from_state_dict(sd) == (scale, nc, kpt)
    d = tmp_path / "yolo"
    d.mkdir()
    assert CK.find_yolo_weights(d) is None and CK.find_yolo_weights(tmp_path / "nope") is None
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(d / "cow.safetensors"))
    f = CK.find_yolo_weights(d)
    cfg2, sd2 = CK.load_yolo_weights(f)
    assert (cfg2.scale, cfg2.nc, cfg2.kpt_shape) == (scale, nc, kpt)
    assert all(np.array_equal(sd2[k], sd[k]) for k in sd)
    # `YOLO(p).model.state_dict()` saved with torch.save, names prefixed by the wrapper ("model.model.N...") + BN counters
    pt = d / "cow.pt"
    wrapped = {"model." + k: torch.from_numpy(v) for k, v in sd.items()}
    wrapped["model.model.1.bn.num_batches_tracked"] = torch.tensor(7)
    torch.save(wrapped, pt)
    assert CK.find_yolo_weights(d).name == "cow.pt"  # *.pt first, like the service's glob
    cfg3, sd3 = CK.load_yolo_weights(pt)
    assert cfg3.scale == scale and all(np.array_equal(sd3[k], sd[k]) for k in sd)


class _Pickled:  # stands for the DetectionModel object an Ultralytics .pt pickles
    pass


def test_stock_ultralytics_pt_is_refused_without_executing_it(tmp_path):
    p = tmp_path / "yolov8n.pt"
    torch.save({"model": _Pickled(), "epoch": -1}, p)
    with pytest.raises(RuntimeError, match="pickles the model object"):
        CK.load_yolo_weights(p)


@pytest.mark.parametrize("arch", ["dinov2", "dinov3"])
def test_dino_model_dir_round_trip(tmp_path, arch):
    from safetensors.numpy import save_file

    if arch == "dinov2":
        cfg = dino.DinoConfig(arch="dinov2", hidden=64, layers=2, heads=2, mlp=256, patch=14, registers=0, eps=1e-6, pos_grid=37)
        hf = {"model_type": "dinov2", "hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 2, "mlp_ratio": 4,
              "patch_size": 14, "image_size": 518, "layer_norm_eps": 1e-6}
    else:
        cfg = dino.DinoConfig(hidden=64, layers=2, heads=2, mlp=192, registers=4)
        hf = {"model_type": "dinov3_vit", "hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 2,
              "intermediate_size": 192, "patch_size": 16, "num_register_tokens": 4, "layer_norm_eps": 1e-5, "rope_theta": 100.0}
    sd = weights.synth_state_dict(dino.param_spec(cfg), 8)
    (tmp_path / "config.json").write_text(json.dumps(hf))
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    cfg2, sd2 = CK.load_dino_dir(tmp_path)
    assert cfg2 == cfg
    assert all(np.array_equal(sd2[k], sd[k]) for k in sd)
    os.remove(tmp_path / "model.safetensors")
    save_file({k: np.ascontiguousarray(v) for k, v in list(sd.items())[:-2]}, str(tmp_path / "model.safetensors"))
    with pytest.raises(RuntimeError, match="tensors missing"):
        CK.load_dino_dir(tmp_path)
