"""Host-side YOLO logic (no GPU): the restated architecture reproduces Ultralytics' published parameter counts and
GFLOPs exactly; letterbox geometry / cv2-linear tables / scale_boxes known answers; lmx's launch-plan bookkeeping
(lmx.yolo) and the oracle's independent yaml walk (oracle.yolo) agree on every tensor name and folded weight."""
import os

import numpy as np
import pytest
import torch

from lmx import letterbox as LB
from lmx import synth, yolo
from oracle import yolo as OY

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("scale,params,gflops", [("n", 3157200, 8.7), ("s", 11166560, 28.6), ("m", 25902640, 78.9),
                                                  ("l", 43691520, 165.2), ("x", 68229648, 257.8)])
def test_published_params_and_flops(scale, params, gflops):
    p, macs = yolo.count_params_flops(yolo.YoloConfig(scale))
    assert p == params
    assert abs(2 * macs / 1e9 - gflops) < 0.06


def test_letterbox_geometry_known_answers():
    g = LB.geometry(1080, 1920)
    assert (g.rh, g.rw, g.top, g.left, g.oh, g.ow) == (360, 640, 12, 0, 384, 640)
    assert (g.pad_x, g.pad_y) == (0.0, 12.0) and abs(g.gain - 1 / 3) < 1e-12
    g = LB.geometry(640, 640)
    assert (g.rh, g.rw, g.top, g.left, g.oh, g.ow) == (640, 640, 0, 0, 640, 640)
    g = LB.geometry(720, 1280)
    assert (g.rh, g.rw, g.oh, g.ow, g.top) == (360, 640, 384, 640, 12)
    g = LB.geometry(1000, 600)  # portrait: width padded to a multiple of 32
    assert (g.rh, g.rw) == (640, 384) and g.ow == 384 and g.oh == 640
    assert OY.letterbox_geometry(1080, 1920)[:3] == (360, 640, 12)


@pytest.mark.parametrize("h,w", [(1080, 1920), (720, 1280), (480, 500), (300, 200)])
def test_letterbox_restatements_agree(h, w):
    f = synth.synth_frame(11, 3, h, w)
    a = LB.letterbox_reference(f, LB.geometry(h, w), swap_rb=False)
    b = OY.letterbox(f)
    assert a.shape == b.shape and np.array_equal(a, b)


def test_cv2_linear_known_properties():
    # constant image stays constant; exact 2x decimation of a horizontal ramp averages neighbours
    img = np.full((64, 64, 3), 77, np.uint8)
    assert np.all(OY.cv2_resize_linear_u8(img, 20, 30) == 77)
    ramp = np.tile(np.arange(64, dtype=np.uint8)[None, :, None] * 4, (8, 1, 3))
    out = OY.cv2_resize_linear_u8(ramp, 32, 8)
    assert np.array_equal(out[0, :, 0], (ramp[0, 0::2, 0].astype(int) + ramp[0, 1::2, 0]) // 2)


def test_scale_boxes_known_answer():
    b = OY.scale_boxes((384, 640), np.array([[10, 12, 630, 372], [-5, 0, 700, 400]], np.float32), (1080, 1920))
    assert np.allclose(b[0], [30, 0, 1890, 1080]) and np.allclose(b[1], [0, 0, 1920, 1080])


def test_fold_and_names_agree_with_oracle():
    cfg = yolo.YoloConfig("n")
    sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("n"))
    # every name the oracle touches exists, and the oracle's own fold equals lmx's
    x = torch.rand(1, 3, 64, 64)
    seen = []
    orig = OY._fused_conv

    def spy(sd_, name, xx, k, s, act=True):
        seen.append(name)
        w, b = yolo.fold_bn(sd_, name)
        g, be = torch.from_numpy(sd_[name + ".bn.weight"]), torch.from_numpy(sd_[name + ".bn.bias"])
        mu, var = torch.from_numpy(sd_[name + ".bn.running_mean"]), torch.from_numpy(sd_[name + ".bn.running_var"])
        sc = g / torch.sqrt(var + 1e-3)
        assert np.allclose(w, (torch.from_numpy(sd_[name + ".conv.weight"]) * sc.view(-1, 1, 1, 1)).numpy(), atol=1e-6)
        assert np.allclose(b, (be - mu * sc).numpy(), atol=1e-6)
        return orig(sd_, name, xx, k, s, act)

    OY._fused_conv = spy
    try:
        with torch.no_grad():
            out = OY.model_forward("n", 80, sd, x)
    finally:
        OY._fused_conv = orig
    assert out.shape == (1, 84, 4 + 16 + 64)  # 64x64 input -> 8x8 + 4x4 + 2x2 anchors
    conv_names = {k[:-len(".conv.weight")] for k in yolo.param_spec(cfg) if k.endswith(".conv.weight")}
    assert set(seen) == conv_names


def test_oracle_detections_match_committed_golden():
    g = np.load(os.path.join(GOLD, "yolov8n_det_w7.npz"))
    cfg = yolo.YoloConfig("n")
    sd = yolo.synthetic_state_dict(cfg, int(g["weight_seed"]), yolo.bn_stats_path("n"))
    cs, fi = g["frames"][2]
    r = OY.predict("n", 80, sd, synth.synth_frame(int(cs), int(fi)), conf=0.5)
    assert np.array_equal(r["src"], g["f2_c50_src"]) and np.array_equal(r["cls"], g["f2_c50_cls"])
    assert np.allclose(r["boxes"], g["f2_c50_boxes"], atol=1e-3) and np.allclose(r["scores"], g["f2_c50_scores"], atol=1e-6)


# ------------------------------------------------------------------------------------- YOLOv8-pose (tleap consumer)
@pytest.mark.parametrize("scale,params,gflops", [("n", 3295470, 9.2), ("s", 11626046, 30.2), ("m", 26464462, 81.0),
                                                  ("l", 44489196, 168.6), ("x", 69491724, 263.2)])
def test_pose_published_params_and_flops(scale, params, gflops):
    """Pose head (cv4, c4 = max(ch0 // 4, 51), nc = 1) against Ultralytics' published yolov8{n..x}-pose figures
    (3.3 / 11.6 / 26.4 / 44.4 / 69.4 M parameters, 9.2 / 30.2 / 81.0 / 168.6 / 263.2 GFLOPs; n: 3,295,470 exactly)."""
    p, macs = yolo.count_params_flops(yolo.YoloConfig(scale, nc=1, kpt_shape=(17, 3)))
    assert abs(p - params) <= (0 if scale == "n" else 60000)  # only the n figure is published to the unit
    assert round(p / 1e5) == round(params / 1e5)
    assert abs(2 * macs / 1e9 - gflops) < 0.15


def test_pose_decode_and_scale_coords_known_answers():
    """kpts_decode on a hand-made head output and scale_coords on the 1080p letterbox (gain 1/3, pad (0, 12) UNROUNDED)."""
    cfg = yolo.YoloConfig("n", nc=1, kpt_shape=(2, 3))
    sd = yolo.synthetic_state_dict(cfg, 1)
    # zero the pose branch's last conv and set its bias: every anchor then predicts raw (0.5, -0.25, 0) for keypoint 0
    for l in range(3):
        sd[f"model.22.cv4.{l}.2.weight"] = np.zeros_like(sd[f"model.22.cv4.{l}.2.weight"])
        sd[f"model.22.cv4.{l}.2.bias"] = np.asarray([0.5, -0.25, 0.0, 0.0, 0.0, 2.0], np.float32)
    x = torch.zeros((1, 3, 64, 96))
    with torch.no_grad():
        y = OY.model_forward("n", 1, sd, x, kpt_shape=(2, 3))[0]
    kp = y[5:].T.reshape(-1, 2, 3).numpy()  # [A, 2, 3]
    # level 0 (stride 8, 8 x 12 cells): anchor 13 = cell (y 1, x 1): x = (0.5*2 + 1) * 8 = 16, y = (-0.25*2 + 1) * 8 = 4
    assert np.allclose(kp[13, 0], [16.0, 4.0, 0.5])
    assert np.allclose(kp[13, 1], [8.0, 8.0, 1 / (1 + np.exp(-2.0))])
    # level 1 starts at anchor 96 (stride 16, 4 x 6 cells): anchor 96 + 7 = cell (1, 1)
    assert np.allclose(kp[96 + 7, 0], [(1.0 + 1) * 16, (-0.5 + 1) * 16, 0.5])
    c = OY.scale_coords((384, 640), np.asarray([[[320.0, 12.0, 0.7], [700.0, 500.0, 0.1]]], np.float32), (1080, 1920))
    assert np.allclose(c[0, 0], [960.0, 0.0, 0.7]) and np.allclose(c[0, 1], [1920.0, 1080.0, 0.1])


def test_pose_oracle_matches_committed_golden():
    cfg = yolo.YoloConfig("n", nc=1, kpt_shape=(17, 3))
    sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("n", pose=True))
    gold = np.load(os.path.join(GOLD, "yolov8n-pose_det_w7.npz"))
    cs, fi = gold["frames"][0]
    r = OY.predict_pose("n", 1, (17, 3), sd, synth.synth_frame(int(cs), int(fi)), conf=float(gold["conf"]))
    assert np.array_equal(r["src"], gold["f0_src"])
    # (fp32 convolutions sum in a thread-count-dependent order: a host with another core count moves a keypoint by ~1e-3 px)
    assert np.allclose(r["keypoints"], gold["f0_keypoints"], atol=2e-2)
    assert np.allclose(r["kpt_raw"][::97], gold["f0_kpt_raw_sample"], atol=2e-3)
