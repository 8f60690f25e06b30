"""The three service mirrors over the real HIP backends on a short synthetic 1080p clip: the JSON files and NATS payloads
of one `video.preprocessed` message, compared with what the fp32 oracle computes for the same sampled frames
(BASELINE cfg#1/#5 semantics, reference schedule: YOLO/SAM every fps//2-th frame, DINO every fps-th)."""
import asyncio
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fused_service_end_to_end(cuda, tmp_path):
    from lmx import dino, sam, sam_decoder, services, synth, weights, yolo
    from lmx.services import runtime as R
    from lmx.services import sam3_pipeline as SP
    from oracle import hiera as OH
    from oracle import preprocess as OP
    from oracle import sam_decoder as OD
    from oracle import vit as OV
    from oracle import yolo as OY

    frames = np.stack([synth.synth_frame(3, 40 + 3 * i) for i in range(9)], 0)
    clip = tmp_path / "clip.npz"
    R.save_npz_clip(clip, frames, 8.0)  # YOLO/SAM: frames 0,4,8; DINO: frames 0,8
    # backends: YOLOv8-n, a narrow Hiera (same block structure, 1024^2 input, 256-d FPN) + the SAM decoder, a small DINOv3
    ycfg = yolo.YoloConfig("n")
    ysd = yolo.synthetic_state_dict(ycfg, 7, yolo.bn_stats_path("n"))
    hcfg = sam.HieraConfig(hidden=16, blocks=(1, 2, 3, 2), dims=(16, 32, 64, 128), heads=(1, 2, 4, 8), global_blocks=(4,),
                           pos_bkg=(7, 7), fpn_dim=256, image=1024)
    hsd = weights.synth_state_dict(sam.param_spec(hcfg), 51)
    msd = sam_decoder.synthetic_state_dict(52)
    dcfg = dino.DinoConfig(hidden=256, layers=3, heads=4, mlp=1024, registers=4)
    dsd = weights.synth_state_dict(dino.param_spec(dcfg), 53)
    bus = R.InProcessBus()
    cfg = {"nats": {"subjects": dict(R.DEFAULT_SUBJECTS)}, "models": {"yolo": {"confidence_threshold": 0.5}}}
    y = services.YOLOPipeline(yolo.YoloDetector(ycfg, ysd, cuda), bus, cfg, results_dir=tmp_path / "yolo")
    seg = SP.HieraSegmenter(sam.HieraEncoder(hcfg, hsd, cuda), sam_decoder.MaskDecoder(msd, cuda))
    s = services.SAM3Pipeline(seg, bus, cfg, results_dir=tmp_path / "sam3", yolo_results_dir=tmp_path / "yolo")
    d = services.DINOv3Pipeline(dino.DinoEmbedder(dcfg, dsd, cuda), bus, None, cfg, results_dir=tmp_path / "dino")
    msg = {"video_id": "clip1", "processed_path": str(clip), "filename": "clip1.mp4"}
    for svc in (y, s, d):  # the three services one after the other, as the product runs them
        asyncio.run(svc.process_video(msg))
    assert [p[0] for p in bus.published] == ["pipeline.yolo", "pipeline.sam3", "pipeline.dinov3"]
    # the fused service: ONE decode, overlapped uploads, FusedExtractor.step per chunk -> the same three files, byte for byte
    from lmx import pipeline

    bus2 = R.InProcessBus()
    y2 = services.YOLOPipeline(y.yolo_model, bus2, cfg, results_dir=tmp_path / "f_yolo")
    s2 = services.SAM3Pipeline(seg, bus2, cfg, results_dir=tmp_path / "f_sam3", yolo_results_dir=tmp_path / "f_yolo")
    d2 = services.DINOv3Pipeline(d.model, bus2, None, cfg, results_dir=tmp_path / "f_dino")
    fx = pipeline.FusedExtractor.from_models(y.yolo_model, seg.encoder, seg.decoder, d.model)
    for schedule, chunk in (("reference", 2), ("dense", 4)):
        fused = services.FusedFeatureService(fx, y2, s2, d2, schedule=schedule, chunk=chunk)
        bus2.published.clear()
        bus2.handlers.clear()
        asyncio.run(fused.start())
        asyncio.run(bus2.publish("video.preprocessed", msg))
        assert [p[0] for p in bus2.published] == ["video.preprocessed", "pipeline.yolo", "pipeline.sam3", "pipeline.dinov3"]
        for a, b in (("yolo/clip1_yolo.json", "f_yolo/clip1_yolo.json"), ("sam3/clip1_sam3.json", "f_sam3/clip1_sam3.json"),
                     ("dino/clip1_dinov3.json", "f_dino/clip1_dinov3.json")):
            ja, jb = json.load(open(tmp_path / a)), json.load(open(tmp_path / b))
            if "dinov3" in a:  # the second run finds the first run's vector in the store: compare the rest
                ja.pop("similar_cases"), jb.pop("similar_cases"), ja.pop("neighbor_evidence"), jb.pop("neighbor_evidence")
            assert ja == jb, f"fused ({schedule}) and separate services disagree on {a}"

    yj = json.load(open(tmp_path / "yolo" / "clip1_yolo.json"))
    sj = json.load(open(tmp_path / "sam3" / "clip1_sam3.json"))
    dj = json.load(open(tmp_path / "dino" / "clip1_dinov3.json"))
    assert yj["fps"] == 8 and yj["total_frames"] == 9
    got = {fr["frame"]: fr["detections"] for fr in yj["detections"]}
    for fid in (0, 4, 8):
        ref = OY.predict("n", 80, ysd, frames[fid], conf=0.5)
        dets = got.get(fid, [])
        assert abs(len(dets) - len(ref["src"])) <= max(2, len(ref["src"]) // 10), (fid, len(dets), len(ref["src"]))
        if len(ref["src"]) and (len(ref["scores"]) < 2 or ref["scores"][0] - ref["scores"][1] > 1e-2):
            assert dets[0]["class_id"] == int(ref["cls"][0])
            assert np.allclose(dets[0]["bbox"], ref["boxes"][0], atol=3.0)
            assert abs(dets[0]["confidence"] - float(ref["scores"][0])) < 2e-2
    # SAM3: prompted with the service's own first detection; compare mask features with the oracle mask of the same prompt
    assert [sg["frame"] for sg in sj["segmentations"]] == [0, 4, 8]
    for sg in sj["segmentations"]:
        fid = sg["frame"]
        assert sg["mask_available"] == (fid in got)
        if not sg["mask_available"]:
            continue
        box = np.array([got[fid][0]["bbox"]], np.float32)
        with torch.no_grad():
            fpn, _ = OH.encoder_forward(hcfg, hsd, torch.from_numpy(OP.sam_pixel_values(frames[fid], 1024))[None])
            sp = OD.prompt_encode_box(msd, torch.from_numpy(OD.scale_box(box, (1080, 1920), (576, 1024))))
            low, _ = OD.mask_decode(msd, fpn[2], sp)
            mask = OD.postprocess(low, (576, 1024), (1080, 1920))[0].numpy()
        ref = SP.extract_segmentation_features(mask)
        f = sg["features"]
        assert f["frame"] == fid and f["time"] == fid / 8
        assert abs(f["mask_area"] - ref["mask_area"]) <= 0.002 * max(ref["mask_area"], 1) + 50
        assert abs(f["centroid_x"] - ref["centroid_x"]) < 2 and abs(f["centroid_y"] - ref["centroid_y"]) < 2
    # DINOv3: canonical frames = [first, middle, last] of the sampled frames (0, 8)
    assert dj["embedding_dim"] == 256 and dj["num_embeddings"] == 2 and [c["frame"] for c in dj["canonical_frames"]] == [0, 8, 8]
    for c in dj["canonical_frames"]:
        with torch.no_grad():
            ref = OV.embed(dcfg, dsd, torch.from_numpy(OP.dino_pixel_values(frames[c["frame"]]))[None])[0]
        cos = torch.nn.functional.cosine_similarity(torch.tensor(c["embedding"], dtype=torch.float64), ref.double(), dim=0)
        assert float(cos) > 1 - 1e-4


def test_tleap_pose_estimator_surface(cuda):
    """services/tleap-pipeline/app/main.py:152-171: per detection a bbox, a confidence and name-keyed model keypoints."""
    import os

    import numpy as np
    import torch

    from lmx import synth, yolo
    from lmx.services import PoseEstimator
    from lmx.services.pose import KEYPOINT_NAMES

    gold = yolo.bn_stats_path("n", pose=True)
    cfg = yolo.YoloConfig("n", nc=1, kpt_shape=(17, 3))
    det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, gold), cuda)
    est = PoseEstimator(det, conf=0.05)
    frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40), synth.synth_frame(2, 50)], 0)).to(cuda)
    res = est.detect_with_trained_model(frames)
    assert len(res) == 2 and all(len(r) > 0 for r in res)
    d = res[0][0]
    assert set(d) == {"bbox", "confidence", "model_keypoints"} and len(d["bbox"]) == 4
    assert list(d["model_keypoints"]) == KEYPOINT_NAMES[:17]
    kp = d["model_keypoints"]["withers"]
    assert set(kp) == {"name", "x", "y", "confidence"} and 0 <= kp["x"] <= 1920 and 0 <= kp["y"] <= 1080 and 0 < kp["confidence"] < 1


def test_clip_curation_tracker(cuda):
    """track_cow_through_video's per-frame records (clip-curation main.py:154-165) over a batch of frames, yolov8n."""
    import os

    import numpy as np
    import torch

    from lmx import synth, yolo
    from lmx.services import CowTracker
    from lmx.services.curation import best_detection

    gold = yolo.bn_stats_path("n")
    cfg = yolo.YoloConfig("n")
    det = yolo.YoloDetector(cfg, yolo.synthetic_state_dict(cfg, 7, gold), cuda)
    frames = torch.from_numpy(np.stack([synth.synth_frame(3, i) for i in (40, 41, 42)], 0)).to(cuda)
    recs = CowTracker(det, conf=0.3, batch=2).track(frames, fps=30.0, first_frame=100)
    assert [r["frame"] for r in recs] == [100, 101, 102] and abs(recs[1]["time"] - 101 / 30.0) < 1e-12
    # same batches as the tracker: the GEMM dispatch depends on the row count (the 12x20 level of a 2-frame batch has
    # M = 480 < 512 rows and takes the register-staged kernel), so another batching may differ in the last f32 bit of an
    # accumulation — and a random 60-layer network amplifies that to a fraction of a pixel
    for i0 in (0, 2):
        boxes, scores, cls, _, counts = (t.cpu().numpy() for t in det.detect(frames[i0:i0 + 2], conf=0.3))
        for b in range(boxes.shape[0]):
            r = recs[i0 + b]
            ref = best_detection(boxes[b], scores[b], cls[b], counts[b], 1080, 1920)
            assert (r["detection"] is None) == (ref is None)
            if ref is not None:
                assert r["detection"]["bbox"] == ref["bbox"] and r["detection"]["area"] == ref["area"]
    assert any(r["detection"] is not None for r in recs)


@pytest.mark.parametrize("n_frames", [20, 64])
def test_fused_step_is_bit_reproducible(cuda, n_frames):
    """FusedExtractor.step on several HIP streams, on one stream, and repeated: identical bits for every output field.
    64 frames is the bench batch (four SAM chunks dealt over the stream pool), 20 a ragged one."""
    import numpy as np
    import torch

    from lmx import pipeline, synth

    fx = pipeline.FusedExtractor(cuda)
    frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(n_frames)], 0)).to(cuda)

    def run(serial):
        fx.serial = serial
        out = fx.step(frames, keep_byte_masks=True)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in out.items()}

    ref = run(True)
    for serial in (False, False, False, True, False):
        got = run(serial)
        for k in ref:
            assert torch.equal(got[k], ref[k]), f"{k} differs between runs (serial={serial})"
    fx.serial = False


@pytest.mark.parametrize("precision", [None, "f16"])
def test_fused_step_on_shards_equals_the_whole_clip(cuda, precision):
    """What makes the multi-GPU JSON equal to the single-GPU one: FusedExtractor.step on contiguous blocks of a
    clip — the ranks' shards, ragged here: 13 + 11 of 24 frames — returns, frame for frame, the bits of the step on the whole clip,
    on the exact plans the services run and on the throughput plans; every kernel choice is independent of the batch size."""
    import numpy as np
    import torch

    from lmx import pipeline, synth

    fx = pipeline.FusedExtractor(cuda)
    frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(24)], 0)).to(cuda)
    whole = {k: v.clone() for k, v in fx.step(frames, keep_byte_masks=True, precision=precision).items()}
    parts = [{k: v.clone() for k, v in fx.step(frames[a:b], keep_byte_masks=True, precision=precision).items()} for a, b in ((0, 13), (13, 24))]
    torch.cuda.synchronize()
    for k, v in whole.items():
        if v.dim() == 0 or v.shape[0] != 24:
            continue
        got = torch.cat([p[k] for p in parts], 0)
        assert torch.equal(got, v), f"{k}: the shards' rows differ from the whole clip's"
    assert any(v.dim() > 0 and v.shape[0] == 24 for v in whole.values())
