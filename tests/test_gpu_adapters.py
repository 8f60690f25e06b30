"""The Python seam driven with the reference's own call idioms — the lines quoted are the glue code of
services/{yolo,sam3,dinov3,tleap}-pipeline/app/main.py with `self.` dropped — against the direct liblmx calls and the
fp32 oracle.  Checkpoints are synthetic files written in the reference's namings (segment_anything .pth, Ultralytics
`model.N.*` safetensors, a Hugging Face model directory) and loaded through the same selection rules as the services."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_yolo_adapter_reference_idiom(cuda, tmp_path):
    from safetensors.numpy import save_file

    from lmx import adapters, checkpoints, synth, yolo

    cfg = yolo.YoloConfig("n")
    sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("n"))
    d = tmp_path / "models" / "yolo"
    d.mkdir(parents=True)
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(d / "cow_detector.safetensors"))
    # yolo main.py:24-30
    model_path = d
    model_file = checkpoints.find_yolo_weights(model_path)
    yolo_model = adapters.LmxYolo(str(model_file), device=cuda)
    confidence_threshold = 0.25
    frame = synth.synth_frame(3, 40)
    # yolo main.py:76-96, verbatim loop
    results = yolo_model(frame, verbose=False, conf=confidence_threshold)
    frame_detections = []
    for result in results:
        boxes = result.boxes
        for box in boxes:
            x1, y1, x2, y2 = box.xyxy[0].cpu().numpy()
            conf = float(box.conf[0].cpu().numpy())
            cls = int(box.cls[0].cpu().numpy())
            class_name = yolo_model.names[cls] if hasattr(yolo_model, "names") else f"class_{cls}"
            frame_detections.append({"bbox": [float(x1), float(y1), float(x2), float(y2)], "confidence": conf, "class": class_name,
                                     "class_id": cls})
    det = yolo.YoloDetector(cfg, sd, cuda)
    b, s, c, _, n = det.detect(torch.from_numpy(frame[None]).to(cuda), conf=confidence_threshold)
    k = int(n[0])
    assert k == len(frame_detections) and k > 0
    for j, dct in enumerate(frame_detections):
        assert dct["bbox"] == [float(v) for v in b[0, j].cpu().numpy()]
        assert dct["confidence"] == float(s[0, j]) and dct["class_id"] == int(c[0, j]) and dct["class"] == yolo.COCO_NAMES[dct["class_id"]]
    json.dumps(frame_detections)  # what the service writes must be JSON-serialisable
    # the caller's buffer may be reused after the call (cv2 does): results stay what they were
    before = results[0].boxes.xyxy.clone()
    frame[:] = 0
    torch.cuda.synchronize()
    assert torch.equal(results[0].boxes.xyxy, before)
    # empty result: iteration yields nothing
    res0 = yolo_model(np.zeros((480, 640, 3), np.uint8), verbose=False, conf=0.99)
    assert len(res0) == 1 and len(res0[0].boxes) == 0 and list(res0[0].boxes) == []


def test_yolo_pose_adapter_tleap_idiom(cuda):
    from lmx import adapters, synth, yolo

    cfg = yolo.YoloConfig("n", nc=1, kpt_shape=(17, 3))
    sd = yolo.synthetic_state_dict(cfg, 7, yolo.bn_stats_path("n", pose=True))
    model = adapters.LmxYolo((cfg, sd), device=cuda)
    frame = synth.synth_frame(3, 40)
    # tleap main.py:142-163
    results = model(frame, verbose=False, conf=0.05)
    got = []
    for result in results:
        if result.boxes is None or len(result.boxes) == 0:
            continue
        for j, box in enumerate(result.boxes):
            bbox = box.xyxy[0].cpu().numpy().tolist()
            confidence = float(box.conf[0].cpu().numpy())
            kpts = None
            if result.keypoints is not None and j < len(result.keypoints):
                kpts = result.keypoints[j].data[0].cpu().numpy()
            got.append((bbox, confidence, kpts))
    det = yolo.YoloDetector(cfg, sd, cuda)
    b, s, c, _, n, kp = det.detect_pose(torch.from_numpy(frame[None]).to(cuda), conf=0.05)
    assert len(got) == int(n[0]) > 0
    for j, (bbox, confidence, kpts) in enumerate(got):
        assert bbox == b[0, j].cpu().numpy().tolist() and confidence == float(s[0, j])
        assert kpts.shape == (17, 3) and np.array_equal(kpts, kp[0, j].cpu().numpy())


def test_sam_adapter_reference_idiom(cuda, tmp_path):
    from lmx import adapters, checkpoints, sam, sam_decoder, synth, weights
    from oracle import preprocess as OP
    from oracle import sam_decoder as OD
    from oracle import sam_vit as OV

    cfg = sam.SamVitConfig(hidden=128, layers=3, heads=2, mlp=256, global_idx=(1,), window=14, image=1024)
    sd = weights.synth_state_dict(sam.vit_param_spec(cfg), 61)
    sd.update(sam_decoder.synthetic_state_dict(62))
    d = tmp_path / "models" / "sam3"
    d.mkdir(parents=True)
    # sam3 main.py:51-56: no checkpoint -> predictor stays None (rectangle fallback in the service)
    assert checkpoints.find_sam_checkpoint(d) == (None, None)
    ckpt = d / "sam_vit_b_synthetic.pth"
    torch.save({k: torch.from_numpy(v) for k, v in checkpoints.lmx_to_segment_anything(sd).items()}, ckpt)
    # sam3 main.py:54-66 (the registry builds vit_b's real size; the synthetic file is smaller, so bypass the size check)
    checkpoint_file, model_type = checkpoints.find_sam_checkpoint(d)
    assert model_type == "vit_b"
    with pytest.raises(RuntimeError, match="not a vit_b"):
        adapters.sam_model_registry[model_type](checkpoint=str(checkpoint_file))
    cfg2, sd2 = checkpoints.load_sam_checkpoint(checkpoint_file)
    sam_model = adapters.LmxSam(cfg2, sd2, cuda)
    sam_predictor = adapters.SamPredictor(sam_model)
    image = synth.synth_frame(6, 20)
    bbox = [420.0, 360.0, 1010.0, 850.0]
    # sam3 main.py:80-89, verbatim
    sam_predictor.set_image(image)
    x1, y1, x2, y2 = bbox
    box = np.array([x1, y1, x2, y2])
    masks, scores, _ = sam_predictor.predict(point_coords=None, point_labels=None, box=box[None, :], multimask_output=False)
    mask = masks[0]
    assert mask.dtype == bool and mask.shape == image.shape[:2] and scores.shape == (1,) and _.shape == (1, 256, 256)
    # against the fp32 oracle from the same raw frame
    pv = torch.from_numpy(OP.sam_pixel_values(image, 1024))[None]
    with torch.no_grad():
        emb = OV.encoder_forward(cfg, sd, pv)
        rhw = sam.resize_longest_side(1080, 1920, 1024)
        sp = OD.prompt_encode_box(sd, torch.from_numpy(OD.scale_box(np.asarray([bbox], np.float32), (1080, 1920), rhw)))
        low, iou = OD.mask_decode(sd, emb, sp)
        ref = OD.postprocess(low, rhw, (1080, 1920))[0].numpy()
    inter, union = float((mask & ref).sum()), float((mask | ref).sum())
    print("sam adapter: IoU vs oracle", inter / union, "coverage", ref.mean(), "iou head", scores, iou.tolist())
    assert 0.02 < ref.mean() < 0.98 and inter / union >= 0.995
    assert abs(float(scores[0]) - float(iou[0])) < 5e-3
    # set_image -> predict is the only state: a second prompt on the cached embedding needs no new set_image
    m2, _, _ = sam_predictor.predict(box=np.array([100.0, 100.0, 900.0, 700.0]), multimask_output=False)
    assert m2.shape == masks.shape and (m2 != masks).any()
    with pytest.raises(RuntimeError):
        adapters.SamPredictor(sam_model).predict(box=box, multimask_output=False)


def test_dino_adapter_reference_idiom(cuda, tmp_path):
    from PIL import Image
    from safetensors.numpy import save_file

    from lmx import adapters, dino, synth, weights
    from oracle import preprocess as OP
    from oracle import vit

    cfg = dino.DinoConfig(arch="dinov2", hidden=192, layers=3, heads=3, mlp=768, patch=14, registers=0, eps=1e-6, pos_grid=37)
    sd = weights.synth_state_dict(dino.param_spec(cfg), 22)
    (tmp_path / "config.json").write_text(json.dumps({"model_type": "dinov2", "hidden_size": 192, "num_hidden_layers": 3,
                                                      "num_attention_heads": 3, "mlp_ratio": 4, "patch_size": 14, "image_size": 518,
                                                      "layer_norm_eps": 1e-6}))
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    device = cuda
    # dinov3 main.py:34-36
    model = adapters.LmxDinoModel.from_pretrained(str(tmp_path)).to(device)
    processor = adapters.LmxImageProcessor.from_pretrained(model)
    model.eval()
    image = synth.synth_frame(10, 3, 720, 1280)
    # dinov3 main.py:98-113 (cv2.cvtColor(image, COLOR_BGR2RGB) written as the slice it is)
    image_rgb = np.ascontiguousarray(image[:, :, ::-1])
    pil_image = Image.fromarray(image_rgb)
    inputs = processor(images=pil_image, return_tensors="pt").to(device)
    with torch.no_grad():
        outputs = model(**inputs)
        embedding = outputs.last_hidden_state.mean(dim=1).squeeze().cpu().numpy()
    assert outputs.last_hidden_state.shape == (1, cfg.tokens, cfg.hidden) and embedding.shape == (192,)
    pv = torch.from_numpy(OP.dino_pixel_values(image))[None]
    with torch.no_grad():
        ref = vit.embed(cfg, sd, pv)[0].numpy()
    cos = float(np.dot(embedding, ref) / (np.linalg.norm(embedding) * np.linalg.norm(ref)))
    assert cos > 1 - 1e-4, cos
    # the adapter path and the batched service path agree to f32 summation order
    direct = dino.DinoEmbedder(cfg, sd, cuda).embed_frames(torch.from_numpy(image[None]).to(cuda))[0].cpu().numpy()
    assert np.allclose(direct, embedding, atol=1e-5)
    # dummy forward for the embedding size (dinov3 main.py:74-78 builds a dummy image for _ensure_collection)
    dummy = processor(images=Image.new("RGB", (224, 224)), return_tensors="pt").to(device)
    assert model(**dummy).last_hidden_state.mean(dim=1).shape[-1] == 192


def test_single_frame_calls_replay_hip_graphs_with_identical_results(cuda, monkeypatch):
    """lmx/graphs.py (LMX_GRAPHS=1): the adapters capture a single-frame call once per input shape and replay it.  Over a sequence of different
    frames (and prompts) the replayed results are byte-identical to the eager calls (LMX_GRAPHS=0), the capture happened exactly
    once, and a result handed out earlier is not overwritten by a later call."""
    from PIL import Image

    from lmx import adapters, dino, sam, sam_decoder, synth, weights, yolo

    frames = [synth.synth_frame(3 + i, 40 - 7 * i) for i in range(3)]
    ycfg = yolo.YoloConfig("n")
    ymodel = adapters.LmxYolo((ycfg, yolo.synthetic_state_dict(ycfg, 7, yolo.bn_stats_path("n"))), device=cuda)
    scfg = sam.SamVitConfig(hidden=128, layers=3, heads=2, mlp=256, global_idx=(1,), window=14, image=1024)
    ssd = weights.synth_state_dict(sam.vit_param_spec(scfg), 61)
    ssd.update(sam_decoder.synthetic_state_dict(62))
    pred = adapters.SamPredictor(adapters.LmxSam(scfg, ssd, cuda))
    dcfg = dino.DinoConfig(arch="dinov2", hidden=192, layers=3, heads=3, mlp=768, patch=14, registers=0, eps=1e-6, pos_grid=37)
    dmodel = adapters.LmxDinoModel(dcfg, weights.synth_state_dict(dino.param_spec(dcfg), 22), cuda)
    proc = adapters.LmxImageProcessor(dmodel)
    boxes = [np.array([420.0, 360.0, 1010.0, 850.0]), np.array([100.0, 100.0, 900.0, 700.0]), np.array([800.0, 200.0, 1700.0, 1000.0])]

    def run_all():
        out = []
        for fr, bx in zip(frames, boxes):
            r = ymodel(fr, verbose=False, conf=0.25)[0]
            pred.set_image(fr)
            m, s, low = pred.predict(box=bx[None, :], multimask_output=False)
            h = dmodel(**proc(images=Image.fromarray(np.ascontiguousarray(fr[:, :, ::-1])), return_tensors="pt").to(cuda)).last_hidden_state
            out.append((r.boxes.xyxy, r.boxes.conf, r.boxes.cls, m, s, low, h))
        return [[t.cpu().numpy() if isinstance(t, torch.Tensor) else t for t in row] for row in out]  # read AFTER the last call

    monkeypatch.setenv("LMX_GRAPHS", "1")
    graphed = run_all()
    fns = [next(iter(ymodel._graphs.values())), pred._g_encode, next(iter(pred._g_decode.values())), dmodel._g_hidden, dmodel._g_pre]
    assert all(not g.failed and len(g.cache) == 1 for g in fns), [(g.failed, len(g.cache)) for g in fns]
    monkeypatch.setenv("LMX_GRAPHS", "0")
    eager = run_all()
    assert len(graphed[0][0]) > 0 and any(m[3].any() for m in graphed)
    for a, b in zip(graphed, eager):
        for x, y in zip(a, b):
            assert x.shape == y.shape and x.dtype == y.dtype and x.tobytes() == y.tobytes()
    assert not np.array_equal(graphed[0][6], graphed[1][6])  # the frames differ, and so do their results
