"""lmx.yolo's two launch plans walked on the CPU with stand-in kernels (tests/cpu_kernels.py): buffer slicing, the x3
channel-group layout of the exact plan and its weight packing.  The exact plan must reproduce the fp32 oracle
(oracle/yolo.py; yolo main.py:76) to fp32 rounding level, the f16 plan to f16 level.  The arithmetic itself is checked on
the device by tests/test_gpu_yolo.py."""
import os

import numpy as np
import pytest
import torch

import cpu_kernels

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(monkeypatch, cfg, sd, img, precision):
    from lmx import yolo

    cpu_kernels.install(monkeypatch)
    det = yolo.YoloDetector(cfg, sd, "cpu", precision=precision)
    return det.forward_letterboxed(torch.from_numpy(img))


@pytest.mark.parametrize("scale,pose", [("n", False), ("n", True), ("s", False)])
def test_exact_plan_matches_fp32_oracle(monkeypatch, scale, pose):
    from lmx import yolo
    from oracle import yolo as OY

    kshape = (17, 3) if pose else None
    cfg = yolo.YoloConfig(scale, nc=1 if pose else 80, kpt_shape=kshape)
    bn = yolo.bn_stats_path(scale, pose=pose)
    sd = yolo.synthetic_state_dict(cfg, 7, bn if os.path.exists(bn) else None)
    img = np.random.default_rng(3).integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
    x = torch.from_numpy(img).permute(0, 3, 1, 2).float() / 255
    if not os.path.exists(bn):  # no committed BatchNorm statistics for this scale: calibrate on the test batch (oracle/yolo.py)
        sd.update(OY.calibrate_bn(scale, cfg.nc, sd, x, kpt_shape=kshape))
    with torch.no_grad():
        ref = OY.model_forward(scale, cfg.nc, sd, x, kpt_shape=kshape).transpose(1, 2).numpy()
    got = _run(monkeypatch, cfg, sd, img, "exact")
    if pose:
        got, kraw = got
        assert len(kraw) == 3 and kraw[0].dtype == torch.float32
    got = got.numpy()
    ref = ref[..., :4 + cfg.nc]
    dbox = np.abs(got[..., :4] - ref[..., :4]).max()
    dcls = np.abs(got[..., 4:] - ref[..., 4:]).max()
    print(f"exact plan vs fp32 oracle (yolov8{scale}{' pose' if pose else ''}): box {dbox:.2e} px, scores {dcls:.2e}")
    assert dbox < 5e-3 and dcls < 2e-5  # fp32 rounding level (the oracle itself is ~1.4e-6 / 3e-4 px from an f64 evaluation)
    f16 = _run(monkeypatch, cfg, sd, img, "f16")
    f16 = (f16[0] if pose else f16).numpy()
    assert np.abs(f16[..., 4:] - ref[..., 4:]).max() < 3e-2 and np.abs(f16[..., 4:] - ref[..., 4:]).max() > 10 * dcls


def test_split_rows_x3_reconstructs_the_weights():
    from lmx import yolo

    rng = np.random.default_rng(0)
    w = (rng.standard_normal((24, 40)) * np.exp(rng.uniform(-12, 2, (24, 1)))).astype(np.float32)
    w[3] = 0
    w[5, ::3] *= 1e-4
    groups = [16, 24]
    x3, sc, e = yolo.split_rows_x3(w, groups)
    assert x3.shape == (24, 120) and x3.dtype == np.float16 and np.isfinite(x3.astype(np.float32)).all()
    o, rec = 0, []
    for g in groups:
        hi, mid, lo = (x3[:, 3 * o + k * g:3 * o + (k + 1) * g].astype(np.float64) for k in range(3))
        assert np.array_equal(mid * 2048, hi) or np.abs(mid * 2048 - hi).max() <= np.abs(hi).max() * 2.0 ** -10
        rec.append(hi + lo)
        o += g
    rec = np.concatenate(rec, 1) * sc[:, None].astype(np.float64)
    amax = np.abs(w).max(1, keepdims=True)
    assert (np.abs(rec - w) <= np.maximum(np.abs(w) * 2.0 ** -21, amax * 2.0 ** -36) + 1e-45).all()
    assert np.all(np.abs(x3.astype(np.float32)).max(1)[np.arange(24) != 3] > 2.0 ** 12)
