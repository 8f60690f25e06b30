"""GPU parity of the YOLO path (services/yolo-pipeline/app/main.py:76) through the C-ABI.
  * byte/index kernels (letterbox, max-pool, upsample, NMS on a given prediction tensor): bit-exact;
  * EXACT plan (lmx.yolo precision="exact": what the services, the adapters and the reference schedule run): the prediction
    tensor agrees with the fp32 oracle to fp32 rounding level (scores <= 1e-5) and the NMS keep-sets / box indices / classes
    are IDENTICAL to the committed fp32 goldens — np.array_equal, no margin rule (north_star: bit-exact keep-sets);
  * F16 plan (the dense THROUGHPUT schedule only): f16 activations deviate 3e-3 .. 6e-3 in score from fp32, so its keep-sets
    are held to the margin rule of tests/keepset.py (borderline candidates counted and bounded) — labelled as such."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import keepset as KS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# caps on the measured deviation of the f16 network from the fp32 oracle near the thresholds (synthetic random weights,
# ~60 layers of f16 activation storage); the tests print the measured values, the margin rule uses the measured ones
EPS_SCORE_CAP, EPS_IOU_CAP = 7.5e-3, 8e-3  # the f16 plan's error budget; measured on MI355X: 3.0e-3 .. 6.4e-3 and 3.3e-3 .. 6.9e-3


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("h,w", [(1080, 1920), (720, 1280), (480, 500), (384, 640)])
def test_letterbox_bit_exact(cuda, h, w):
    from lmx import kernels as K
    from lmx import letterbox as LB
    from lmx import synth
    from oracle import yolo as OY

    frames = np.stack([synth.synth_frame(12, i, h, w) for i in (0, 5)], 0)
    frames[1] = np.random.default_rng(1).integers(0, 256, frames[1].shape, dtype=np.uint8)
    geo = LB.geometry(h, w)
    tabs = None
    if (geo.rh, geo.rw) != (h, w):
        tabs = tuple(torch.from_numpy(t).to(cuda) for t in LB.resize_tables(h, w, geo.rh, geo.rw))
    out = K.letterbox(torch.from_numpy(frames).to(cuda), geo, tabs, swap_rb=True).cpu().numpy()
    for i in range(2):
        ref = OY.letterbox(frames[i])[:, :, ::-1]
        assert np.array_equal(out[i], ref), f"{h}x{w} frame {i}"


def test_stem_pool_upsample_decode(cuda):
    from lmx import kernels as K

    # stem conv
    img = np.random.default_rng(2).integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
    w = _rand((16, 3, 3, 3), 3, 0.3)
    b = _rand((16,), 4, 0.1)
    x = torch.from_numpy(img).permute(0, 3, 1, 2).float() / 255
    ref = F.silu(F.conv2d(x, w, b, stride=2, padding=1)).permute(0, 2, 3, 1)
    got = K.stem_conv(torch.from_numpy(img).to(cuda), w.permute(2, 3, 1, 0).contiguous().to(cuda), b.to(cuda))
    assert float((got.float().cpu() - ref).abs().max()) < 2e-3
    # max pool 5 on slices + upsample
    buf = _rand((2, 20, 12, 64), 5).half()
    d = buf.to(cuda)
    K.maxpool5(d[..., :16], d[..., 16:32])
    ref = F.max_pool2d(buf[..., :16].float().permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1)
    assert torch.equal(d[..., 16:32].float().cpu(), ref)
    assert torch.equal(d[..., 32:].cpu(), buf[..., 32:])
    up = torch.zeros((2, 40, 24, 48), dtype=torch.float16, device=cuda)
    K.upsample2(d[..., 16:32], up[..., 8:24])
    refu = F.interpolate(ref.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(up[..., 8:24].float().cpu(), refu) and float(up[..., :8].abs().max()) == 0
    # detect decode
    n, H, W, nc = 2, 6, 10, 80
    head = _rand((n, H, W, 64 + nc), 6, 2.0)
    pred = torch.zeros((n, H * W + 7, 4 + nc), dtype=torch.float32, device=cuda)
    K.detect_decode(head.to(cuda), pred, nc, 16.0, 7)
    box = head[..., :64].reshape(n, H * W, 4, 16).softmax(-1)
    dist = (box * torch.arange(16.0)).sum(-1)
    gy, gx = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    a = torch.stack((gx, gy), -1).view(-1, 2)
    x1y1, x2y2 = a - dist[..., :2], a + dist[..., 2:]
    refp = torch.cat(((x1y1 + x2y2) / 2 * 16, (x2y2 - x1y1) * 16, head[..., 64:].reshape(n, H * W, nc).sigmoid()), -1)
    got = pred[:, 7:].cpu()
    assert float((got - refp).abs().max()) < 1e-4
    assert float(pred[:, :7].abs().max()) == 0


@pytest.mark.parametrize("scale,frames", [("n", [(3, 40), (2, 50), (4, 0)]), ("l", [(3, 40), (2, 50)])])
def test_yolo_f16_plan_margin_rule(cuda, scale, frames):
    """The dense throughput plan (precision="f16") against the fp32 oracle: margin rule, NOT the drop-in parity bar (that is
    test_yolo_exact_plan_keepsets_equal_fp32 below)."""
    from lmx import kernels as K
    from lmx import synth, yolo
    from oracle import nms as ONMS
    from oracle import yolo as OY

    cfg = yolo.YoloConfig(scale)
    sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path(scale))
    det = yolo.YoloDetector(cfg, sd, cuda, precision="f16")
    gold = np.load(os.path.join(GOLD, f"yolov8{scale}_det_w7.npz"))
    fr = np.stack([synth.synth_frame(cs, fi) for cs, fi in frames], 0)
    d_fr = torch.from_numpy(fr).to(cuda)
    img, geo = det.preprocess(d_fr)
    pred = det.forward_letterboxed(img)
    torch.cuda.synchronize()
    pred_c = pred.cpu().numpy()
    report = []
    for j, (cs, fi) in enumerate(frames):
        ref = OY.predict(scale, cfg.nc, sd, fr[j], conf=0.5)
        emu = OY.predict(scale, cfg.nc, sd, fr[j], conf=0.5, emulate_f16=True)
        # (1a) vs the fp32 arithmetic with f16 STORAGE emulated.  Accumulation order differs, and every value that
        # sits near a half-precision rounding boundary then rounds the other way, so this is NOT tighter than the
        # quantisation noise itself — it bounds the kernels to that noise level (a wrong kernel is off by O(1)).
        ebox = np.abs(pred_c[j][:, :4] - emu["pred"][:, :4]).max()
        ecls = np.abs(pred_c[j][:, 4:] - emu["pred"][:, 4:]).max()
        assert ebox < 3.0 and ecls < 8e-3, f"frame {j}: kernels disagree with f16-storage emulation: box {ebox} cls {ecls}"
        # (1b) vs the plain fp32 oracle: the price of f16 activations through ~60 layers, bounded
        dbox = np.abs(pred_c[j][:, :4] - ref["pred"][:, :4]).max()
        dcls = np.abs(pred_c[j][:, 4:] - ref["pred"][:, 4:]).max()
        assert dbox < 4.0, f"frame {j}: box coords off by {dbox} px (letterboxed)"
        assert dcls < 2e-2, f"frame {j}: class scores off by {dcls}"
        report.append(("pred", j, float(ebox), float(ecls), float(dbox), float(dcls)))
        ref_nms = {c: ONMS.non_max_suppression(ref["pred"], c) for c in (0.25, 0.5)}
        assert np.allclose(ref["pred"][::97], gold[f"f{j}_pred_sample"], atol=1e-4), "oracle drifted from golden"
        for conf in (0.25, 0.5):
            # (2) NMS on the GPU's own prediction tensor: bit-exact against the oracle NMS
            b, s, c, src, cnt = (t.cpu().numpy() for t in K.nms(pred[j:j + 1].contiguous(), conf))
            rb, rs, rc, rsrc = ONMS.non_max_suppression(pred_c[j], conf)
            k = len(rsrc)
            assert cnt[0] == k and np.array_equal(src[0, :k], rsrc) and np.array_equal(c[0, :k], rc)
            assert np.array_equal(b[0, :k], rb) and np.array_equal(s[0, :k], rs)
            # (3) end to end vs the fp32 oracle: the keep-set (anchor indices = box indices) must be IDENTICAL to the fp32
            # one after removing only the candidates the margin rule lists as ambiguous at the MEASURED deviation of this
            # frame's device prediction (tests/keepset.py); any unexplained difference fails
            g_src = gold[f"f{j}_c{int(conf * 100)}_src"]
            assert np.array_equal(ref_nms[conf][3], g_src), "oracle keep-set drifted from the committed golden"
            cref, cdev = KS.compact_pred(ref["pred"]), KS.compact_pred(pred_c[j])
            eps_s, eps_i, n_meas = KS.measure_eps(cref, cdev, conf)
            assert eps_s <= EPS_SCORE_CAP and eps_i <= EPS_IOU_CAP, f"frame {j} conf {conf}: f16 deviation eps_score {eps_s} eps_iou {eps_i}"
            rep = KS.check_keepset(cref, conf, 0.7, eps_s, eps_i, rsrc, rc, f"yolov8{scale} frame {j} conf {conf}")
            report.append((j, conf, rep["n_dev"], len(g_src), rep["n_firm"], rep["n_ambiguous"], float(eps_s), float(eps_i)))
            # the margin must leave the rule something to check (at conf 0.25 these synthetic weights give 500+ NMS survivors,
            # so the max_det = 300 cut adds rank ambiguity on top: the floor there is lower)
            assert rep["n_firm"] >= (0.5 if len(g_src) < 300 else 0.25) * len(g_src) or len(g_src) < 4, f"margin too wide to mean anything: {rep} vs {len(g_src)} fp32 detections"
            if len(g_src) and k:
                # the detection SAM is prompted with (first = highest confidence) must be the same anchor unless
                # the fp32 top-2 scores are within 2 * eps_score of each other
                gs = gold[f"f{j}_c{int(conf * 100)}_scores"]
                if len(gs) < 2 or gs[0] - gs[1] > 2 * eps_s:
                    assert rsrc[0] == g_src[0], f"frame {j}: top-1 detection differs"
    print("yolo", scale, "(frame, conf, kept_gpu, kept_fp32, firm, ambiguous, eps_score, eps_iou):", report)
    # full detect() path incl. scale_boxes on the batch
    boxes, scores, cls, src, counts = det.detect(d_fr, conf=0.5)
    torch.cuda.synchronize()
    for j in range(len(frames)):
        k = int(counts[j])
        rb, rs, rc, rsrc = ONMS.non_max_suppression(pred_c[j], 0.5)
        assert k == len(rsrc)
        if k:
            want = OY.scale_boxes((geo.oh, geo.ow), rb, fr[j].shape[:2])
            assert np.allclose(boxes[j, :k].cpu().numpy(), want, atol=1e-3)
            assert float(boxes[j, :k, [0, 2]].max()) <= 1920 and float(boxes[j, :k, [1, 3]].max()) <= 1080


def test_pose_gather_kernel_exact(cuda):
    """lmx_k_pose_gather on hand-made level outputs: decode + unrounded-pad scale_coords + clip + sigmoid, zeros past counts."""
    from lmx import kernels as K
    from oracle import yolo as OY

    rng = np.random.default_rng(40)
    n, kshape, ldk = 2, (5, 3), 16
    dims = [(6, 10), (3, 5), (2, 3)]
    raws = [rng.standard_normal((n, h, w, ldk)).astype(np.float32) for h, w in dims]
    A = sum(h * w for h, w in dims)
    src = np.full((n, 8), -1, np.int32)
    src[0, :5] = [0, 59, 60, 74, A - 1]
    src[1, :2] = [7, 61]
    counts = np.asarray([5, 2], np.int32)
    gain, padx, pady, fw, fh = 0.25, 1.5, 10.25, 300.0, 120.0
    got = K.pose_gather([torch.from_numpy(r).to(cuda) for r in raws], (8, 16, 32), torch.from_numpy(src).to(cuda),
                        torch.from_numpy(counts).to(cuda), kshape, padx, pady, gain, fw, fh).cpu().numpy()
    flat = np.concatenate([r.reshape(n, -1, ldk) for r in raws], 1)  # [n, A, ldk]
    cell = np.concatenate([np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2) for h, w in dims], 0)
    strd = np.concatenate([np.full(h * w, s, np.float32) for (h, w), s in zip(dims, (8, 16, 32))])
    for b in range(n):
        for j in range(8):
            if j >= counts[b]:
                assert not got[b, j].any()
                continue
            a = src[b, j]
            v = flat[b, a, :15].reshape(5, 3)
            x = (v[:, 0] * np.float32(2) + cell[a, 0].astype(np.float32)) * strd[a]
            y = (v[:, 1] * np.float32(2) + cell[a, 1].astype(np.float32)) * strd[a]
            ref = np.clip(np.stack([(x - np.float32(padx)) / np.float32(gain), (y - np.float32(pady)) / np.float32(gain)], -1),
                          0, [fw, fh]).astype(np.float32)
            assert np.array_equal(got[b, j, :, :2], ref), (b, j)
            assert np.allclose(got[b, j, :, 2], 1 / (1 + np.exp(-v[:, 2])), atol=2e-7)


def test_pose_end_to_end(cuda):
    """YOLOv8n-pose (services/tleap-pipeline/app/main.py:142-163: boxes + `result.keypoints[j].data`) against the fp32
    oracle on 1080p frames.  As for detection the conv stack is f16: raw keypoint offsets must agree to quantisation
    level, and detections kept by both paths must carry the same keypoints to a fraction of a pixel."""
    from lmx import synth, yolo
    from oracle import yolo as OY

    kshape = (17, 3)
    cfg = yolo.YoloConfig("n", nc=1, kpt_shape=kshape)
    sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("n", pose=True))
    det = yolo.YoloDetector(cfg, sd, cuda, precision="f16")
    gold = np.load(os.path.join(GOLD, "yolov8n-pose_det_w7.npz"))
    conf = float(gold["conf"])
    fr = np.stack([synth.synth_frame(int(cs), int(fi)) for cs, fi in gold["frames"]], 0)
    d_fr = torch.from_numpy(fr).to(cuda)
    boxes, scores, cls, src, counts, kpts = (t.cpu().numpy() for t in det.detect_pose(d_fr, conf=conf))
    img, geo = det.preprocess(d_fr)
    pred, kraw = det.forward_letterboxed(img)
    raw = torch.cat([k.reshape(k.shape[0], -1, k.shape[-1]) for k in kraw], 1).cpu().numpy()[..., :51]  # [n, A, 51]
    for j in range(fr.shape[0]):
        ref = OY.predict_pose("n", 1, kshape, sd, fr[j], conf=conf)
        assert np.array_equal(ref["src"], gold[f"f{j}_src"]), "oracle drifted from golden"
        # raw keypoint head vs the oracle's decoded tensor, undone: x_raw = (x_dec / stride - cell) / 2
        k = int(counts[j])
        common = sorted(set(src[j, :k].tolist()) & set(ref["src"].tolist()))
        assert len(common) >= 0.8 * max(k, len(ref["src"])), f"frame {j}: keep-sets differ too much ({len(common)} of {k}/{len(ref['src'])})"
        gi = {a: i for i, a in enumerate(src[j, :k].tolist())}
        ri = {a: i for i, a in enumerate(ref["src"].tolist())}
        dxy = max(float(np.abs(kpts[j, gi[a], :, :2] - ref["keypoints"][ri[a], :, :2]).max()) for a in common)
        dv = max(float(np.abs(kpts[j, gi[a], :, 2] - ref["keypoints"][ri[a], :, 2]).max()) for a in common)
        dbox = max(float(np.abs(boxes[j, gi[a]] - ref["boxes"][ri[a]]).max()) for a in common)
        print(f"pose frame {j}: {k} detections ({len(common)} common), keypoint xy max diff {dxy:.3f} px, visibility {dv:.4f}, box {dbox:.3f} px")
        assert dxy < 6.0 and dv < 2e-2 and dbox < 12.0  # frame pixels = 3x letterboxed pixels
        # f16-storage emulation: bounds the kernels themselves
        emu = OY.predict_pose("n", 1, kshape, sd, fr[j], conf=conf, emulate_f16=True)
        sel = np.asarray(common[:64])
        lv = np.where(sel >= 80 * 48 + 40 * 24, 2, np.where(sel >= 80 * 48, 1, 0))  # 384x640 letterbox: 48x80, 24x40, 12x20
        strd = np.asarray([8.0, 16.0, 32.0], np.float32)[lv]
        dec = emu["kpt_raw"][sel].reshape(len(sel), 17, 3)
        got_raw = raw[j, sel].reshape(len(sel), 17, 3)
        # compare decoded-x differences in units of raw offsets: d(x_dec) = 2 * stride * d(raw)
        offs = np.concatenate([[0], [48 * 80], [48 * 80 + 24 * 40]])
        loc = sel - offs[lv]
        wl = np.asarray([80, 40, 20])[lv]
        cx, cy = (loc % wl).astype(np.float32), (loc // wl).astype(np.float32)
        gx = (got_raw[..., 0] * 2 + cx[:, None]) * strd[:, None]
        gy = (got_raw[..., 1] * 2 + cy[:, None]) * strd[:, None]
        e = max(float(np.abs(gx - dec[..., 0]).max()), float(np.abs(gy - dec[..., 1]).max()))
        assert e < 1.5, f"frame {j}: keypoint head disagrees with the f16-storage emulation by {e} letterboxed px"


def test_cfg2_yolov8l_640x640_batch32(cuda):
    """BASELINE cfg#2 on the throughput (f16) plan: YOLOv8-l on [32,640,640,3] (no letterbox padding: 640x640 frames map 1:1).  The two golden frames
    (indices 0 and 31 of the batch) pass the keep-set margin rule against the committed fp32 oracle prediction; NMS on the
    device prediction is bit-exact against the oracle NMS for every frame of the batch; a second run of the whole batch is
    bit-identical; frames of the batch do not influence each other (frame 31 alone == frame 31 in the batch)."""
    from lmx import kernels as K
    from lmx import synth, yolo
    from oracle import nms as ONMS

    g = np.load(os.path.join(GOLD, "yolov8l_cfg2_w7.npz"))
    cfg = yolo.YoloConfig("l")
    sd = yolo.synthetic_state_dict(cfg, int(g["weight_seed"]), yolo.bn_stats_path("l"))
    det = yolo.YoloDetector(cfg, sd, cuda, precision="f16")
    fr = synth.cfg2_frames()
    assert fr.shape == (32, 640, 640, 3)
    d_fr = torch.from_numpy(fr).to(cuda)
    img, geo = det.preprocess(d_fr)
    assert tuple(img.shape) == (32, 640, 640, 3) and (geo.pad_x, geo.pad_y) == (0, 0)
    assert torch.equal(img, d_fr.flip(-1)), "640x640 input: letterbox must be the identity (BGR->RGB only)"
    pred = det.forward_letterboxed(img)
    torch.cuda.synchronize()
    assert tuple(pred.shape) == (32, 8400, 84)
    pred_c = pred.cpu().numpy()
    assert np.isfinite(pred_c).all()
    report = []
    for conf in (0.25, 0.5):
        b, s, c, src, cnt = (t.cpu().numpy() for t in K.nms(pred, conf))
        for i in range(32):  # NMS kernel vs oracle NMS on the same (device) prediction: bit-exact, every frame
            rb, rs, rc, rsrc = ONMS.non_max_suppression(pred_c[i], conf)
            k = len(rsrc)
            assert cnt[i] == k and np.array_equal(src[i, :k], rsrc) and np.array_equal(c[i, :k], rc), f"frame {i} conf {conf}"
            assert np.array_equal(b[i, :k], rb) and np.array_equal(s[i, :k], rs)
        for j, fi in enumerate(g["frame_ids"]):  # the golden frames vs the fp32 oracle
            cref = dict(box=g[f"f{j}_box"], score=g[f"f{j}_score"], cls=g[f"f{j}_cls"], score2=g[f"f{j}_score2"])
            cdev = KS.compact_pred(pred_c[fi])
            eps_s, eps_i, _ = KS.measure_eps(cref, cdev, conf)
            assert eps_s <= EPS_SCORE_CAP and eps_i <= EPS_IOU_CAP, f"frame {fi}: eps_score {eps_s} eps_iou {eps_i}"
            k = int(cnt[fi])
            rep = KS.check_keepset(cref, conf, 0.7, eps_s, eps_i, src[fi, :k], c[fi, :k], f"cfg2 frame {fi} conf {conf}")
            n_gold = len(g[f"f{j}_c{int(conf * 100)}_src"])
            report.append((int(fi), conf, k, n_gold, rep["n_firm"], rep["n_ambiguous"], float(eps_s), float(eps_i)))
            assert rep["n_firm"] >= (0.5 if n_gold < 300 else 0.25) * n_gold, f"margin too wide to mean anything: {rep} vs {n_gold}"
    print("cfg2 (frame, conf, kept_gpu, kept_fp32, firm, ambiguous, eps_score, eps_iou):", report)
    # bit-reproducible, and batch-independent
    pred2 = det.forward_letterboxed(img)
    torch.cuda.synchronize()
    assert torch.equal(pred, pred2), "cfg#2 forward is not bit-reproducible"
    alone = det.forward_letterboxed(img[31:32])
    assert torch.equal(alone[0], pred[31]), "frame 31 alone differs from frame 31 inside the batch of 32"
    # the service-level call on the batch
    boxes, scores, cls, src2, counts = det.detect(d_fr, conf=0.5)
    torch.cuda.synchronize()
    b, s, c, src, cnt = (t.cpu().numpy() for t in K.nms(pred, 0.5))
    assert np.array_equal(counts.cpu().numpy(), cnt) and np.array_equal(src2.cpu().numpy(), src)
    assert np.allclose(boxes.cpu().numpy(), np.clip(b, 0, 640), atol=1e-4)  # gain 1, no padding: scale_boxes only clips


# ---- the EXACT plan: keep-sets identical to the fp32 goldens (north_star) -------------------------------------------------------
EXACT_SCORE_EPS = 2e-5   # measured on MI355X: <= 1.01e-5 over every kept detection (yolov8l), 4e-6 on the prediction sample; the fp32
                         # oracle itself sits 1.4e-6 from an f64 evaluation of yolov8n
EXACT_BOX_EPS = 5e-3     # letterboxed px on coordinates up to 640 (f32 ulp there: 6e-5)


FP32_TIE = 5e-6  # two fp32 scores closer than this have no defined order in the REFERENCE: the fp32 CPU path sits 1e-6 .. 4e-6 from an
#                  f64 evaluation of itself (tools/fp32_noise_probe.py, profiles/r03_fp32_noise.txt) and reorders with the core count


def _assert_keepset_equals_golden(det_out, j, gold, prefix, conf_key, scale_px=1.0):
    """The device keep-set IS the fp32 golden's: same count, same anchors, same order (np.array_equal).  The one thing that is
    not a property of the fp32 path itself is the order of two detections whose fp32 scores differ by less than the fp32 path's
    own rounding noise (FP32_TIE): such neighbours may appear swapped; they are counted, printed and must stay rare.  Nothing
    else is tolerated: no missing / extra anchor, no class change, no swap across a larger score gap."""
    boxes, scores, cls, src, counts = det_out
    g_src, g_sc = gold[f"{prefix}src"], gold[f"{prefix}scores"]
    k = int(counts[j])
    assert k == len(g_src), f"{prefix}: {k} detections, fp32 golden has {len(g_src)}"
    dev = src[j, :k]
    ties = 0
    perm = np.arange(k)
    if not np.array_equal(dev, g_src):
        assert sorted(dev.tolist()) == sorted(g_src.tolist()), f"{prefix}: kept anchors differ from the fp32 golden's"
        where = {int(a): i for i, a in enumerate(g_src)}
        perm = np.asarray([where[int(a)] for a in dev])
        moved = np.nonzero(perm != np.arange(k))[0]
        gap = float(np.abs(g_sc[perm[moved]] - g_sc[moved]).max())
        assert gap <= FP32_TIE, f"{prefix}: detections {moved.tolist()} are ordered differently across an fp32 score gap of {gap}"
        ties = len(moved)
        assert ties <= 4, f"{prefix}: {ties} detections sit in fp32 score ties — too many to call the keep-set pinned"
    if f"{prefix}cls" in gold.files:
        assert np.array_equal(cls[j, :k], gold[f"{prefix}cls"][perm]), f"{prefix}: classes differ"
    ds = float(np.abs(scores[j, :k] - g_sc[perm]).max()) if k else 0.0
    db = float(np.abs(boxes[j, :k] - gold[f"{prefix}boxes"][perm]).max()) if k else 0.0
    assert ds <= EXACT_SCORE_EPS and db <= EXACT_BOX_EPS * scale_px, f"{prefix}: scores off by {ds}, boxes by {db} px"
    return k, ties, ds, db, perm


@pytest.mark.parametrize("scale,frames", [("n", [(3, 40), (2, 50), (4, 0)]), ("l", [(3, 40), (2, 50)])])
def test_yolo_exact_plan_keepsets_equal_fp32(cuda, scale, frames):
    """precision="exact" from raw 1080p frames: detect() returns EXACTLY the fp32 oracle's keep-set (anchor indices in NMS
    order), classes and count at conf 0.25 and 0.5 for every golden frame; scores within 1e-5, frame-pixel boxes within
    1.5e-2 px (gain 1/3).  The raw prediction tensor is compared with the committed sample of the oracle's."""
    from lmx import synth, yolo

    cfg = yolo.YoloConfig(scale)
    sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path(scale))
    det = yolo.YoloDetector(cfg, sd, cuda)  # default plan = exact
    assert det.precision == "exact"
    gold = np.load(os.path.join(GOLD, f"yolov8{scale}_det_w7.npz"))
    fr = np.stack([synth.synth_frame(cs, fi) for cs, fi in frames], 0)
    d_fr = torch.from_numpy(fr).to(cuda)
    img, geo = det.preprocess(d_fr)
    pred = det.forward_letterboxed(img).cpu().numpy()
    report = []
    for j in range(len(frames)):
        ref = gold[f"f{j}_pred_sample"]  # rows [::97] of the fp32 oracle's [A, 84]
        got = pred[j][::97]
        eps_s, eps_b = float(np.abs(got[:, 4:] - ref[:, 4:]).max()), float(np.abs(got[:, :4] - ref[:, :4]).max())
        assert eps_s <= EXACT_SCORE_EPS and eps_b <= EXACT_BOX_EPS, f"frame {j}: prediction off by {eps_s} (scores) / {eps_b} px"
        report.append(("pred", j, eps_s, eps_b))
    for conf in (0.25, 0.5):
        out = tuple(t.cpu().numpy() for t in det.detect(d_fr, conf=conf))
        for j in range(len(frames)):
            report.append((j, conf) + _assert_keepset_equals_golden(out, j, gold, f"f{j}_c{int(conf * 100)}_", conf, 3.0)[:4])
    print(f"yolov8{scale} exact plan (frame, conf, kept = fp32 kept, detections in an fp32 score tie, max score diff, max box diff px):", report)
    # the plan is batch-independent and reproducible like the f16 one
    alone = det.forward_letterboxed(img[1:2]).cpu().numpy()
    assert np.array_equal(alone[0], pred[1]), "exact plan: frame alone differs from the frame inside the batch"


def test_cfg2_exact_plan_golden_frames(cuda):
    """BASELINE cfg#2's two golden frames (640 x 640, indices 0 and 31 of the batch) on the exact plan: keep-sets identical
    to the fp32 goldens at conf 0.25 and 0.5, and eps measured against the oracle's full prediction (all 8400 anchors)."""
    from lmx import synth, yolo

    g = np.load(os.path.join(GOLD, "yolov8l_cfg2_w7.npz"))
    cfg = yolo.YoloConfig("l")
    det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, int(g["weight_seed"]), yolo.bn_stats_path("l")), cuda)
    fr = synth.cfg2_frames()[[int(i) for i in g["frame_ids"]]]
    d_fr = torch.from_numpy(fr).to(cuda)
    img, _ = det.preprocess(d_fr)
    pred = det.forward_letterboxed(img).cpu().numpy()
    report = []
    for j in range(2):
        score = pred[j][:, 4:].max(1)
        eps_s = float(np.abs(score - g[f"f{j}_score"]).max())
        eps_b = float(np.abs(pred[j][:, :4] - g[f"f{j}_box"]).max())
        assert np.array_equal(pred[j][:, 4:].argmax(1), g[f"f{j}_cls"]) or eps_s <= EXACT_SCORE_EPS
        assert eps_s <= EXACT_SCORE_EPS and eps_b <= EXACT_BOX_EPS, f"cfg2 frame {j}: eps_score {eps_s} eps_box {eps_b}"
        report.append(("pred", j, eps_s, eps_b))
    for conf in (0.25, 0.5):
        out = tuple(t.cpu().numpy() for t in det.detect(d_fr, conf=conf))
        for j in range(2):
            report.append((j, conf) + _assert_keepset_equals_golden(out, j, g, f"f{j}_c{int(conf * 100)}_", conf)[:4])
    print("cfg2 exact plan (frame, conf, kept = fp32 kept, detections in an fp32 score tie, max score diff, max box diff px):", report)


def test_pose_exact_plan_equals_fp32(cuda):
    """YOLOv8n-pose on the exact plan: the kept detections ARE the fp32 oracle's (anchor indices in order), keypoints within
    2e-2 frame px and 1e-5 in visibility."""
    from lmx import synth, yolo

    kshape = (17, 3)
    cfg = yolo.YoloConfig("n", nc=1, kpt_shape=kshape)
    det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("n", pose=True)), cuda)
    gold = np.load(os.path.join(GOLD, "yolov8n-pose_det_w7.npz"))
    conf = float(gold["conf"])
    fr = np.stack([synth.synth_frame(int(cs), int(fi)) for cs, fi in gold["frames"]], 0)
    boxes, scores, cls, src, counts, kpts = (t.cpu().numpy() for t in det.detect_pose(torch.from_numpy(fr).to(cuda), conf=conf))
    for j in range(fr.shape[0]):
        k, ties, ds, db, perm = _assert_keepset_equals_golden((boxes, scores, cls, src, counts), j, gold, f"f{j}_", conf, 3.0)
        gk = gold[f"f{j}_keypoints"][perm]  # (the golden's rows in the device's order: differs only inside an fp32 score tie)
        dk = float(np.abs(kpts[j, :k, :, :2] - gk[..., :2]).max())
        dv = float(np.abs(kpts[j, :k, :, 2] - gk[..., 2]).max())
        print(f"pose exact frame {j}: {k} detections = fp32 ({ties} in an fp32 score tie), scores {ds:.1e}, boxes {db:.1e} px, keypoints {dk:.1e} px, visibility {dv:.1e}")
        assert dk <= 2e-2 and dv <= EXACT_SCORE_EPS
