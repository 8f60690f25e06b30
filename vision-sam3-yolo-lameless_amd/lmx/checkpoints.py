"""Model selection and real-checkpoint key maps — the load paths of the three services:

  * services/yolo-pipeline/app/main.py:24-37     first `*.pt` in /app/shared/models/yolo, else the stock yolov8n
  * services/sam3-pipeline/app/main.py:51-72     first `*.pth` in /app/shared/models/sam3; `vit_h` / `vit_l` in the FILE
                                                 NAME pick the architecture, anything else is vit_b; no file -> None
                                                 (the service then segments with the bbox rectangle)
  * services/dinov3-pipeline/app/main.py:30-36   models.dinov3.model_name, default "facebook/dinov2-base"

Checkpoints are read with loaders that execute nothing from the file (lmx.weights.load_state_dict_file: safetensors or
torch.load(weights_only=True)).  segment_anything's `.pth` files are plain tensor dicts and load as they are; their
parameter names (image_encoder.* / prompt_encoder.* / mask_decoder.*) are mapped onto the names lmx uses (transformers'
SamModel names, which is what the oracle is pinned to).  An Ultralytics `.pt` pickles the model OBJECT: it cannot be read
without the ultralytics package, so the detector takes a state dict exported from it (`model.N.*` names, as
`YOLO(p).model.state_dict()` gives) — see INTEGRATION.md."""
import json
import re
from pathlib import Path

import numpy as np

from . import weights

# ---- SAM v1: segment_anything <-> lmx (transformers SamModel) names ---------------------------------------------------------
_SA_TO_LMX = [
    (r"^image_encoder\.", "vision_encoder."),
    (r"^vision_encoder\.blocks\.", "vision_encoder.layers."),
    (r"^(vision_encoder\.layers\.\d+)\.norm([12])\.", r"\1.layer_norm\2."),
    (r"^vision_encoder\.patch_embed\.proj\.", "vision_encoder.patch_embed.projection."),
    (r"^vision_encoder\.neck\.0\.", "vision_encoder.neck.conv1."),
    (r"^vision_encoder\.neck\.1\.", "vision_encoder.neck.layer_norm1."),
    (r"^vision_encoder\.neck\.2\.", "vision_encoder.neck.conv2."),
    (r"^vision_encoder\.neck\.3\.", "vision_encoder.neck.layer_norm2."),
    (r"^prompt_encoder\.point_embeddings\.", "prompt_encoder.point_embed."),
    (r"^prompt_encoder\.mask_downscaling\.0\.", "prompt_encoder.mask_embed.conv1."),
    (r"^prompt_encoder\.mask_downscaling\.1\.", "prompt_encoder.mask_embed.layer_norm1."),
    (r"^prompt_encoder\.mask_downscaling\.3\.", "prompt_encoder.mask_embed.conv2."),
    (r"^prompt_encoder\.mask_downscaling\.4\.", "prompt_encoder.mask_embed.layer_norm2."),
    (r"^prompt_encoder\.mask_downscaling\.6\.", "prompt_encoder.mask_embed.conv3."),
    (r"^(mask_decoder\.transformer\.layers\.\d+)\.norm([1-4])\.", r"\1.layer_norm\2."),
    (r"^mask_decoder\.transformer\.norm_final_attn\.", "mask_decoder.transformer.layer_norm_final_attn."),
    (r"^mask_decoder\.output_upscaling\.0\.", "mask_decoder.upscale_conv1."),
    (r"^mask_decoder\.output_upscaling\.1\.", "mask_decoder.upscale_layer_norm."),
    (r"^mask_decoder\.output_upscaling\.3\.", "mask_decoder.upscale_conv2."),
    (r"^(mask_decoder\.(?:output_hypernetworks_mlps\.\d+|iou_prediction_head))\.layers\.0\.", r"\1.proj_in."),
    (r"^(mask_decoder\.(?:output_hypernetworks_mlps\.\d+|iou_prediction_head))\.layers\.2\.", r"\1.proj_out."),
    (r"^(mask_decoder\.(?:output_hypernetworks_mlps\.\d+|iou_prediction_head))\.layers\.1\.", r"\1.layers.0."),
]
_PE_SA = "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"
_PE_LMX = ("shared_image_embedding.positional_embedding", "prompt_encoder.shared_embedding.positional_embedding")


def segment_anything_to_lmx(sd):
    """{segment_anything parameter name: array} -> {lmx (transformers SamModel) name: array}.  The single random-Fourier
    matrix of segment_anything's pe_layer is stored under both tied transformers names."""
    out = {}
    for k, v in sd.items():
        if k == _PE_SA:
            for name in _PE_LMX:
                out[name] = v
            continue
        name = k
        for pat, rep in _SA_TO_LMX:
            name = re.sub(pat, rep, name)
        out[name] = v
    return out


def lmx_to_segment_anything(sd):
    """Inverse of segment_anything_to_lmx (used to write checkpoints in the reference's naming, e.g. by the tests)."""
    inv = [
        (r"^(mask_decoder\.(?:output_hypernetworks_mlps\.\d+|iou_prediction_head))\.layers\.0\.", r"\1.layers.1."),
        (r"^(mask_decoder\.(?:output_hypernetworks_mlps\.\d+|iou_prediction_head))\.proj_in\.", r"\1.layers.0."),
        (r"^(mask_decoder\.(?:output_hypernetworks_mlps\.\d+|iou_prediction_head))\.proj_out\.", r"\1.layers.2."),
        (r"^mask_decoder\.upscale_conv1\.", "mask_decoder.output_upscaling.0."),
        (r"^mask_decoder\.upscale_layer_norm\.", "mask_decoder.output_upscaling.1."),
        (r"^mask_decoder\.upscale_conv2\.", "mask_decoder.output_upscaling.3."),
        (r"^mask_decoder\.transformer\.layer_norm_final_attn\.", "mask_decoder.transformer.norm_final_attn."),
        (r"^(mask_decoder\.transformer\.layers\.\d+)\.layer_norm([1-4])\.", r"\1.norm\2."),
        (r"^prompt_encoder\.mask_embed\.conv1\.", "prompt_encoder.mask_downscaling.0."),
        (r"^prompt_encoder\.mask_embed\.layer_norm1\.", "prompt_encoder.mask_downscaling.1."),
        (r"^prompt_encoder\.mask_embed\.conv2\.", "prompt_encoder.mask_downscaling.3."),
        (r"^prompt_encoder\.mask_embed\.layer_norm2\.", "prompt_encoder.mask_downscaling.4."),
        (r"^prompt_encoder\.mask_embed\.conv3\.", "prompt_encoder.mask_downscaling.6."),
        (r"^prompt_encoder\.point_embed\.", "prompt_encoder.point_embeddings."),
        (r"^vision_encoder\.neck\.conv1\.", "vision_encoder.neck.0."),
        (r"^vision_encoder\.neck\.layer_norm1\.", "vision_encoder.neck.1."),
        (r"^vision_encoder\.neck\.conv2\.", "vision_encoder.neck.2."),
        (r"^vision_encoder\.neck\.layer_norm2\.", "vision_encoder.neck.3."),
        (r"^vision_encoder\.patch_embed\.projection\.", "vision_encoder.patch_embed.proj."),
        (r"^(vision_encoder\.layers\.\d+)\.layer_norm([12])\.", r"\1.norm\2."),
        (r"^vision_encoder\.layers\.", "vision_encoder.blocks."),
        (r"^vision_encoder\.", "image_encoder."),
    ]
    out = {}
    for k, v in sd.items():
        if k == _PE_LMX[0]:
            out[_PE_SA] = v
            continue
        if k == _PE_LMX[1]:
            continue
        name = k
        for pat, rep in inv:
            name = re.sub(pat, rep, name)
        out[name] = v
    return out


def sam_model_type(filename):
    """sam3 main.py:57-63: "vit_h" in the name -> vit_h, else "vit_l" -> vit_l, else vit_b."""
    name = Path(filename).name
    if "vit_h" in name:
        return "vit_h"
    if "vit_l" in name:
        return "vit_l"
    return "vit_b"


def find_sam_checkpoint(models_dir="/app/shared/models/sam3"):
    """-> (path, model type) of the checkpoint the service would load, or (None, None) (sam3 main.py:54-56,68-69)."""
    d = Path(models_dir)
    if d.exists():
        files = list(d.glob("*.pth"))
        if files:
            return files[0], sam_model_type(files[0])
    return None, None


def sam_vit_config_from_state_dict(sd):
    """Architecture of a (mapped) SAM v1 state dict, read off the tensor shapes."""
    from . import sam

    D = int(sd["vision_encoder.pos_embed"].shape[-1])
    grid = int(sd["vision_encoder.pos_embed"].shape[1])
    patch = int(sd["vision_encoder.patch_embed.projection.weight"].shape[-1])
    layers = 1 + max(int(m.group(1)) for m in (re.match(r"vision_encoder\.layers\.(\d+)\.", k) for k in sd) if m)
    hd = int(sd["vision_encoder.layers.0.attn.rel_pos_h"].shape[1])
    glob = tuple(i for i in range(layers) if int(sd[f"vision_encoder.layers.{i}.attn.rel_pos_h"].shape[0]) == 2 * grid - 1)
    win = [(int(sd[f"vision_encoder.layers.{i}.attn.rel_pos_h"].shape[0]) + 1) // 2 for i in range(layers) if i not in glob]
    return sam.SamVitConfig(hidden=D, layers=layers, heads=D // hd, mlp=int(sd["vision_encoder.layers.0.mlp.lin1.weight"].shape[0]),
                            global_idx=glob, window=win[0] if win else 14, patch=patch, image=grid * patch,
                            out_ch=int(sd["vision_encoder.neck.conv1.weight"].shape[0]))


def load_sam_checkpoint(path, model_type=None):
    """segment_anything `.pth` (or a safetensors file of the same names) -> (SamVitConfig, lmx-named state dict).
    `model_type` (vit_b / vit_l / vit_h), when given, must agree with the tensors — as `sam_model_registry[type](checkpoint=)`
    fails on a mismatch (load_state_dict is strict)."""
    from . import sam

    raw = weights.load_state_dict_file(path)
    sd = segment_anything_to_lmx(raw) if any(k.startswith("image_encoder.") for k in raw) else raw
    cfg = sam_vit_config_from_state_dict(sd)
    if model_type is not None:
        want = {"vit_b": sam.sam_vit_b(), "vit_l": sam.sam_vit_l(), "vit_h": sam.sam_vit_h(), "default": sam.sam_vit_h()}[model_type]
        if (cfg.hidden, cfg.layers, cfg.heads, tuple(cfg.global_idx)) != (want.hidden, want.layers, want.heads, tuple(want.global_idx)):
            raise RuntimeError(f"checkpoint {path} is not a {model_type} model (hidden {cfg.hidden}, {cfg.layers} layers, "
                               f"{cfg.heads} heads, global blocks {cfg.global_idx})")
    missing = [k for k in sam.vit_param_spec(cfg) if k not in sd]
    if missing:
        raise RuntimeError(f"checkpoint {path}: {len(missing)} image-encoder tensors missing, e.g. {missing[:3]}")
    return cfg, sd


# ---- YOLOv8 ------------------------------------------------------------------------------------------------------------------
def find_yolo_weights(models_dir="/app/shared/models/yolo", patterns=("*.pt", "*.safetensors")):
    """First weight file in the service's model directory (yolo main.py:25-29), or None (-> the stock yolov8n)."""
    d = Path(models_dir)
    if d.exists():
        for pat in patterns:
            files = list(d.glob(pat))
            if files:
                return files[0]
    return None


def yolo_config_from_state_dict(sd):
    """(scale, nc, kpt_shape) of an Ultralytics YOLOv8 detection / pose state dict (`model.N.*` names), from the shapes."""
    from . import yolo

    c0 = int(sd["model.0.conv.weight"].shape[0])
    nb = 1 + max(int(m.group(1)) for m in (re.match(r"model\.2\.m\.(\d+)\.", k) for k in sd) if m)
    c9 = int(sd["model.9.cv2.conv.weight"].shape[0])
    scale = None
    for s in yolo.SCALES:
        cfg = yolo.YoloConfig(s)
        if (cfg.ch(64), cfg.depth(3), cfg.ch(1024)) == (c0, nb, c9):
            scale = s
    if scale is None:
        raise RuntimeError(f"not a YOLOv8 n/s/m/l/x state dict (stem width {c0}, {nb} bottlenecks in model.2, SPPF width {c9})")
    nc = int(sd["model.22.cv3.0.2.weight"].shape[0])
    kpt = None
    if "model.22.cv4.0.2.weight" in sd:
        nk = int(sd["model.22.cv4.0.2.weight"].shape[0])
        kpt = (nk // 3, 3) if nk % 3 == 0 else (nk // 2, 2)
    return scale, nc, kpt


def load_yolo_weights(path):
    """-> (YoloConfig, state dict) from a safetensors / weights_only file holding `model.N.*` tensors (optionally prefixed
    `model.model.N.*`, as `YOLO(p).model.state_dict()` may be saved)."""
    from . import yolo

    try:
        sd = weights.load_state_dict_file(path)
    except Exception as e:  # noqa: BLE001 — typically the pickled DetectionModel of a stock Ultralytics .pt
        raise RuntimeError(f"{path}: cannot be read without executing code from the file ({type(e).__name__}). An Ultralytics "
                           ".pt pickles the model object; export `YOLO(path).model.state_dict()` to safetensors "
                           "(INTEGRATION.md) and put that file in the models directory.") from e
    if any(k.startswith("model.model.") for k in sd):
        sd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.model.")}
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    scale, nc, kpt = yolo_config_from_state_dict(sd)
    cfg = yolo.YoloConfig(scale, nc=nc, kpt_shape=kpt)
    missing = [k for k in yolo.param_spec(cfg) if k not in sd]
    if missing:
        raise RuntimeError(f"{path}: {len(missing)} tensors missing for yolov8{scale}, e.g. {missing[:3]}")
    return cfg, sd


# ---- DINOv2 / DINOv3 ---------------------------------------------------------------------------------------------------------
def load_dino_dir(model_dir):
    """A local Hugging Face model directory (config.json + model.safetensors) -> (DinoConfig, state dict).  The service asks
    the hub by name (dinov3 main.py:34-35); offline deployments point models.dinov3.model_name at such a directory."""
    from . import dino

    d = Path(model_dir)
    with open(d / "config.json") as f:
        c = json.load(f)
    mt = c.get("model_type", "")
    if mt == "dinov2":
        cfg = dino.DinoConfig(arch="dinov2", hidden=c["hidden_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                              mlp=int(c["hidden_size"] * c.get("mlp_ratio", 4)), patch=c["patch_size"], registers=0,
                              eps=c.get("layer_norm_eps", 1e-6), pos_grid=c.get("image_size", 518) // c["patch_size"])
    elif mt == "dinov3_vit":
        cfg = dino.DinoConfig(arch="dinov3", hidden=c["hidden_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                              mlp=c["intermediate_size"], patch=c["patch_size"], registers=c.get("num_register_tokens", 4),
                              eps=c.get("layer_norm_eps", 1e-5), rope_theta=c.get("rope_theta", 100.0))
    else:
        raise RuntimeError(f"{d}: model_type {mt!r} is neither dinov2 nor dinov3_vit")
    sd = weights.load_state_dict_file(str(d / "model.safetensors"))
    missing = [k for k in dino.param_spec(cfg) if k not in sd]
    if missing:
        raise RuntimeError(f"{d}: {len(missing)} tensors missing, e.g. {missing[:3]}")
    return cfg, {k: np.asarray(v, np.float32) for k, v in sd.items()}
