"""Regenerates the committed golden vectors.  Run in the build container (CPU):  python tests/golden/make_golden.py

The reference holds no vectors for this path and its services cannot be imported here (cv2 /
ultralytics / nats missing, §8c), so the vectors come from the third-party model code the services call, as installed
in the container: transformers' DINOv3ViTModel / Dinov2Model (eager attention, fp32, CPU) fed by the PIL image
processor — with the build's deterministic synthetic weights (lmx.weights, seed in the file name) because no
checkpoint exists offline.  Only inputs' seeds and outputs are stored (no weights, no third-party source)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]

from lmx import dino, synth, weights  # noqa: E402
from oracle import preprocess as OP  # noqa: E402
from oracle import vit  # noqa: E402


def hf_dino(cfg, sd):
    if cfg.arch == "dinov3":
        from transformers import DINOv3ViTConfig, DINOv3ViTModel

        c = DINOv3ViTConfig(hidden_size=cfg.hidden, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers,
                            num_attention_heads=cfg.heads, num_register_tokens=cfg.registers, patch_size=cfg.patch,
                            layer_norm_eps=cfg.eps, rope_theta=cfg.rope_theta, image_size=cfg.image,
                            attn_implementation="eager")
        m = DINOv3ViTModel(c)
    else:
        from transformers import Dinov2Config, Dinov2Model

        c = Dinov2Config(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                         mlp_ratio=cfg.mlp // cfg.hidden, patch_size=cfg.patch, image_size=cfg.pos_grid * cfg.patch,
                         layer_norm_eps=cfg.eps, attn_implementation="eager")
        m = Dinov2Model(c)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.eval()


def hf_pixel_values(frames_bgr):
    from PIL import Image
    from transformers.models.bit.image_processing_pil_bit import BitImageProcessorPil

    proc = BitImageProcessorPil(size={"shortest_edge": 256}, crop_size={"height": 224, "width": 224},
                                image_mean=list(OP.IMAGENET_MEAN), image_std=list(OP.IMAGENET_STD), resample=3)
    return torch.cat([proc(images=Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])), return_tensors="pt")["pixel_values"]
                      for f in frames_bgr], 0)


def make_dino(name, cfg, seed, frame_ids, clip_seed):
    sd = weights.synth_state_dict(dino.param_spec(cfg), seed)
    frames = np.stack([synth.synth_frame(clip_seed, i) for i in frame_ids], 0)
    pv = hf_pixel_values(frames)
    with torch.no_grad():
        hs = hf_dino(cfg, sd)(pixel_values=pv).last_hidden_state
        emb = hs.mean(dim=1)  # services/dinov3-pipeline/app/main.py:113
        ours = vit.embed(cfg, sd, pv)
    cos = torch.nn.functional.cosine_similarity(emb, ours, dim=1)
    print(name, "oracle-vs-transformers max abs", float((emb - ours).abs().max()), "min cos", float(cos.min()))
    assert float((emb - ours).abs().max()) < 5e-4
    np.savez_compressed(os.path.join(HERE, name + ".npz"), embedding=emb.numpy(), cls_token=hs[:, 0].numpy(),
                        weight_seed=seed, clip_seed=clip_seed, frame_ids=np.asarray(frame_ids),
                        pixel_checksum=np.asarray([int(pv.double().abs().sum() * 1000)]))


def make_yolo(scale, seed, calib_clip, frames_spec):
    """BN calibration statistics (synthetic-weight hygiene, see oracle.yolo.calibrate_bn) + the fp32 oracle's
    detections on a few 1080p frames.  No ultralytics/cv2 here: these are ORACLE outputs (parity unpinned)."""
    from lmx import yolo
    from oracle import yolo as OY

    cfg = yolo.YoloConfig(scale)
    sd = yolo.synthetic_state_dict(cfg, seed)
    calib = [synth.synth_frame(calib_clip, i) for i in (10, 90)]
    x = torch.stack([torch.from_numpy(np.ascontiguousarray(OY.letterbox(f)[:, :, ::-1].transpose(2, 0, 1))).float() / 255
                     for f in calib])
    stats = OY.calibrate_bn(scale, cfg.nc, sd, x)
    bn_path = yolo.bn_stats_path(scale, seed)
    np.savez(bn_path, **stats)
    sd = yolo.synthetic_state_dict(cfg, seed, bn_path)
    out = {"weight_seed": seed, "frames": np.asarray(frames_spec)}
    for j, (cs, fi) in enumerate(frames_spec):
        for conf in (0.25, 0.5):
            r = OY.predict(scale, cfg.nc, sd, synth.synth_frame(cs, fi), conf=conf)
            tag = f"f{j}_c{int(conf * 100)}"
            out[tag + "_boxes"], out[tag + "_scores"] = r["boxes"], r["scores"]
            out[tag + "_cls"], out[tag + "_src"] = r["cls"], r["src"]
        out[f"f{j}_pred_sample"] = r["pred"][::97].copy()  # every 97th anchor row of the raw prediction
        print(f"yolov8{scale} frame {cs}/{fi}: {len(r['src'])} detections at conf 0.5")
    np.savez_compressed(os.path.join(HERE, f"yolov8{scale}_det_w{seed}.npz"), **out)


def make_yolo_pose(scale, seed, calib_clip, frames_spec, kpt_shape=(17, 3), conf=0.05):
    """YOLOv8-pose (the tleap-pipeline consumer): BN calibration statistics for the pose variant (nc = 1 changes the class
    branch width, cv4 is new) + the fp32 oracle's detections and keypoints.  ORACLE outputs (parity unpinned)."""
    from lmx import yolo
    from oracle import yolo as OY

    cfg = yolo.YoloConfig(scale, nc=1, kpt_shape=kpt_shape)
    sd = yolo.synthetic_state_dict(cfg, seed)
    calib = [synth.synth_frame(calib_clip, i) for i in (10, 90)]
    x = torch.stack([torch.from_numpy(np.ascontiguousarray(OY.letterbox(f)[:, :, ::-1].transpose(2, 0, 1))).float() / 255
                     for f in calib])
    stats = OY.calibrate_bn(scale, cfg.nc, sd, x, kpt_shape=kpt_shape)
    bn_path = yolo.bn_stats_path(scale, seed, pose=True)
    np.savez(bn_path, **stats)
    sd = yolo.synthetic_state_dict(cfg, seed, bn_path)
    out = {"weight_seed": seed, "frames": np.asarray(frames_spec), "conf": conf}
    for j, (cs, fi) in enumerate(frames_spec):
        r = OY.predict_pose(scale, cfg.nc, kpt_shape, sd, synth.synth_frame(cs, fi), conf=conf)
        out[f"f{j}_boxes"], out[f"f{j}_scores"], out[f"f{j}_src"] = r["boxes"], r["scores"], r["src"]
        out[f"f{j}_keypoints"] = r["keypoints"]
        out[f"f{j}_kpt_raw_sample"] = r["kpt_raw"][::97].copy()
        print(f"yolov8{scale}-pose frame {cs}/{fi}: {len(r['src'])} detections at conf {conf}, max score "
              f"{float(r['pred'][:, 4].max()):.3f}")
    np.savez_compressed(os.path.join(HERE, f"yolov8{scale}-pose_det_w{seed}.npz"), **out)


def make_hiera(seed, clip_seed, frame_ids):
    """fp32 oracle (oracle.hiera, pinned to transformers' Sam2VisionModel by tests/test_oracle_hiera.py) on raw frames."""
    from lmx import sam
    from oracle import hiera as OH

    cfg = sam.hiera_b_plus()
    sd = weights.synth_state_dict(sam.param_spec(cfg), seed)
    frames = [synth.synth_frame(clip_seed, i) for i in frame_ids]
    pv = torch.from_numpy(np.stack([OP.sam_pixel_values(f, 1024) for f in frames], 0))
    with torch.no_grad():
        fpn, stages = OH.encoder_forward(cfg, sd, pv)
    np.savez_compressed(os.path.join(HERE, f"hiera_bplus_w{seed}.npz"), weight_seed=seed, clip_seed=clip_seed,
                        frame_ids=np.asarray(frame_ids),  # spatially subsampled, f16: keeps the fixture ~1 MB
                        fpn2=fpn[2].permute(0, 2, 3, 1)[:, ::2, ::2].numpy().astype(np.float16),
                        stage3=stages[3][:, ::2, ::2].numpy().astype(np.float16),
                        fpn0_sub=fpn[0].permute(0, 2, 3, 1)[:, ::16, ::16].numpy().astype(np.float16))
    print("hiera-b+ golden:", [tuple(f.shape) for f in fpn], "rms fpn2", float(fpn[2].pow(2).mean().sqrt()))


def make_cfg1(scale="n", seed=7, conf=0.001):
    """BASELINE cfg#1: services/yolo-pipeline on ONE 640 x 640 frame presented as a 1-frame clip, YOLOv8-n,
    conf lowered to 0.001 so that NMS and the max_det = 300 cut are exercised; the fp32 oracle's kept detections."""
    from lmx import synth, yolo
    from oracle import yolo as OY

    cfg = yolo.YoloConfig(scale)
    sd = yolo.synthetic_state_dict(cfg, seed, yolo.bn_stats_path(scale, seed))
    frame = synth.cfg1_frame()
    r = OY.predict(scale, cfg.nc, sd, frame, conf=conf)
    print(f"cfg1 yolov8{scale}: {len(r['src'])} detections at conf {conf}, scores {r['scores'][:3]} .. {r['scores'][-3:]}")
    np.savez_compressed(os.path.join(HERE, f"cfg1_yolov8{scale}_w{seed}.npz"), weight_seed=seed, conf=conf, boxes=r["boxes"],
                        scores=r["scores"], cls=r["cls"], src=r["src"], pred_sample=r["pred"][::97])


def make_yolo_cfg2(scale="l", seed=7, frame_ids=(0, 31)):
    """fp32 oracle detections for two frames of the cfg#2 batch (YOLOv8-l, 640x640, batch 32).  Weights: seed 7 with the
    committed BatchNorm statistics (SURVEY names seed 1; the calibrated statistics exist for seed 7 — same architecture)."""
    from lmx import yolo
    from oracle import yolo as OY

    cfg = yolo.YoloConfig(scale)
    sd = yolo.synthetic_state_dict(cfg, seed, yolo.bn_stats_path(scale, seed))
    fr = synth.cfg2_frames()
    out = {"weight_seed": seed, "frame_ids": np.asarray(frame_ids)}
    for j, fi in enumerate(frame_ids):
        for conf in (0.25, 0.5):
            r = OY.predict(scale, cfg.nc, sd, fr[fi], conf=conf)
            tag = f"f{j}_c{int(conf * 100)}"
            out[tag + "_boxes"], out[tag + "_scores"] = r["boxes"], r["scores"]
            out[tag + "_cls"], out[tag + "_src"] = r["cls"], r["src"]
        # what the keep-set margin rule (tests/keepset.py) needs of the fp32 prediction [8400, 84]: box, best score + class, runner-up
        cs = r["pred"][:, 4:]
        cls = cs.argmax(1)
        best = cs[np.arange(len(cs)), cls]
        tmp = cs.copy()
        tmp[np.arange(len(cs)), cls] = -np.inf
        out[f"f{j}_box"], out[f"f{j}_score"] = r["pred"][:, :4].astype(np.float32), best.astype(np.float32)
        out[f"f{j}_cls"], out[f"f{j}_score2"] = cls.astype(np.int16), tmp.max(1).astype(np.float32)
        print(f"cfg2 yolov8{scale} frame {fi}: {len(r['src'])} detections at conf 0.5, pred {r['pred'].shape}")
    np.savez_compressed(os.path.join(HERE, f"yolov8{scale}_cfg2_w{seed}.npz"), **out)


SAM_BOXES = np.array([[420.0, 360.0, 1010.0, 850.0], [1100.5, 380.25, 1700.0, 860.0]], np.float32)


def make_sam_masks(kind, seed, clip_seed=6, frame_ids=(20, 100)):
    """Raw 1080p frame -> mask through the fp32 oracle (set_image + predict(box), services/sam3-pipeline/app/main.py:80-88):
    image encoder (Hiera-B+ = BASELINE cfg#3, or SAM v1 ViT-B = the reference's code path), prompt encoder, mask decoder,
    post-processing.  Stored: the bit-packed mask (np.packbits), the low-res logits (f16), the IoU head and the box."""
    from lmx import sam, sam_decoder
    from oracle import sam_decoder as OD

    frames = [synth.synth_frame(clip_seed, i) for i in frame_ids]
    pv = torch.from_numpy(np.stack([OP.sam_pixel_values(f, 1024) for f in frames], 0))
    dsd = sam_decoder.synthetic_state_dict(seed + 100)
    with torch.no_grad():
        if kind == "hiera_bplus":
            from oracle import hiera as OH

            cfg = sam.hiera_b_plus()
            sd = weights.synth_state_dict(sam.param_spec(cfg), seed)
            emb = OH.encoder_forward(cfg, sd, pv)[0][2]
        else:
            from oracle import sam_vit as OV

            cfg = sam.sam_vit_b()
            sd = weights.synth_state_dict(sam.vit_param_spec(cfg), seed)
            emb = OV.encoder_forward(cfg, sd, pv)
        hw = frames[0].shape[:2]
        rhw = sam.resize_longest_side(hw[0], hw[1], 1024)
        sp = OD.prompt_encode_box(dsd, torch.from_numpy(OD.scale_box(SAM_BOXES, hw, rhw)))
        low, iou = OD.mask_decode(dsd, emb, sp)
        mask = OD.postprocess(low, rhw, hw).numpy()
    cov = mask.reshape(len(frames), -1).mean(1)
    print(f"sam mask golden {kind}: coverage {cov.tolist()}, iou head {iou.tolist()}, emb rms {float(emb.pow(2).mean().sqrt()):.3f}")
    np.savez_compressed(os.path.join(HERE, f"sam_mask_{kind}_w{seed}.npz"), weight_seed=seed, clip_seed=clip_seed,
                        frame_ids=np.asarray(frame_ids), boxes=SAM_BOXES, mask_bits=np.packbits(mask, axis=-1),
                        lowres=low.numpy().astype(np.float16), iou=iou.numpy(), coverage=cov,
                        emb_sub=emb.permute(0, 2, 3, 1)[:, ::4, ::4].numpy().astype(np.float16))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hiera":
        make_hiera(5, 6, [20])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg2":
        make_yolo_cfg2()
        make_cfg1()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg1":
        make_cfg1()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sam_masks":
        make_sam_masks("hiera_bplus", 5)
        make_sam_masks("vit_b", 9)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pose":
        make_yolo_pose("n", 7, 2, [(3, 40), (2, 50)])
        sys.exit(0)
    make_yolo("n", 7, 2, [(3, 40), (2, 50), (4, 0)])
    make_yolo("l", 7, 2, [(3, 40), (2, 50)])
    make_yolo_pose("n", 7, 2, [(3, 40), (2, 50)])
    torch.manual_seed(0)
    make_dino("dinov3_vitl16_w3", dino.dinov3_vitl16(), 3, [0, 75, 149], clip_seed=4)
    make_dino("dinov2_base_w4", dino.dinov2_base(), 4, [0, 30], clip_seed=5)
    make_hiera(5, 6, [20])
    make_yolo_cfg2()
    make_cfg1()
    make_sam_masks("hiera_bplus", 5)
    make_sam_masks("vit_b", 9)
