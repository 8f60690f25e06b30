"""SAM3Pipeline — mirror of services/sam3-pipeline/app/main.py: reads the YOLO result file (cache first), prompts with
the FIRST detection of each sampled frame, writes `{id}_sam3.json` and publishes `pipeline.sam3` (Appendix B.2).
`set_image` + `predict(box=...)` (main.py:80-88) are replaced by HieraEncoder.encode + MaskDecoder.predict; the 7 mask
features come from lmx_h_mask_features on the host copy of the mask.  Without a segmenter the service falls back to the
bbox rectangle exactly like the shipped reference (no checkpoint: main.py:68-69,94-100); a per-frame failure of the
segmenter falls back the same way (main.py:90-92)."""
import ctypes as C
import json
import traceback
from pathlib import Path

import numpy as np
import torch

from .. import _lib
from .. import kernels as K
from . import runtime as R

FEATURE_KEYS = ["mask_area", "area_ratio", "circularity", "aspect_ratio", "centroid_x", "centroid_y", "perimeter"]


def extract_segmentation_features(mask):
    """mask: bool/uint8 [h,w] on the HOST -> the 7-key dict of main.py:137-145."""
    m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
    out = (C.c_double * 7)()
    rc = _lib.load().lmx_h_mask_features(m.ctypes.data_as(C.c_void_p), m.shape[0], m.shape[1], C.cast(out, C.c_void_p))
    if rc != 0:
        raise RuntimeError("lmx_h_mask_features failed")
    return {k: float(v) for k, v in zip(FEATURE_KEYS, out)}


def features_from_device(stats_row, contour_row, h, w):
    """The same 7-key dict from what the DEVICE computed: stats_row = mask_post's int64 [8] (area, sum x, sum y, ...),
    contour_row = lmx_k_contour_features' int64 [8] (2*contourArea, unit steps, diagonal steps, bbox, #external contours).
    Same arithmetic, in the same order, as csrc/host_mask.cpp (doubles), so the two agree bit for bit."""
    m00, m10, m01 = float(stats_row[0]), float(stats_row[1]), float(stats_row[2])
    a2, n_unit, n_diag, minx, miny, maxx, maxy, found = (int(v) for v in contour_row)
    out = {"mask_area": m00, "area_ratio": m00 / float(h * w) if h * w > 0 else 0.0}
    if found > 0:
        area = abs(a2) * 0.5
        per = float(n_unit) + float(n_diag) * 1.41421356237309504880
        out["circularity"] = (4.0 * 3.14159265358979323846 * area) / (per * per) if per > 0 else 0.0
        bw, bh = maxx - minx + 1, maxy - miny + 1
        out["aspect_ratio"] = float(bw) / float(bh) if bh > 0 else 0.0
        perimeter = per
    else:
        out["circularity"], out["aspect_ratio"], perimeter = 0.0, 0.0, 0.0
    if m00 != 0:
        out["centroid_x"], out["centroid_y"] = m10 / m00, m01 / m00
    else:
        out["centroid_x"], out["centroid_y"] = w / 2.0, h / 2.0
    out["perimeter"] = perimeter
    return {k: float(out[k]) for k in FEATURE_KEYS}


def fallback_segmentation(shape_hw, bbox):
    h, w = shape_hw
    mask = np.zeros((h, w), dtype=np.uint8)
    x1, y1, x2, y2 = [int(c) for c in bbox]
    mask[y1:y2, x1:x2] = 255
    return mask.astype(bool)


class SAM3Pipeline:
    def __init__(self, segmenter, bus, config=None, results_dir="/app/data/results/sam3", yolo_results_dir="/app/data/results/yolo",
                 batch=16):
        """segmenter: object with .device and .segment(frames_u8_device [n,h,w,3], boxes f32 device [n,4]) -> u8 masks
        [n,h,w] on device, or None (rectangle fallback)."""
        self.config = config or R.load_config()
        self.nats_client = bus
        self.sam_predictor = segmenter
        self.results_dir = Path(results_dir)
        self.results_dir.mkdir(parents=True, exist_ok=True)
        self.yolo_results_dir = Path(yolo_results_dir)
        self.yolo_results_cache = {}
        self.batch = batch

    async def get_yolo_results(self, video_id):
        if video_id in self.yolo_results_cache:
            return self.yolo_results_cache[video_id]
        f = self.yolo_results_dir / f"{video_id}_yolo.json"
        if f.exists():
            with open(f) as fh:
                res = json.load(fh)
            self.yolo_results_cache[video_id] = res
            return res
        return {}

    def _features(self, frames_host, boxes):
        """frames_host uint8 [k,h,w,3] + their boxes -> list of the 7-key feature dicts of extract_segmentation_features.
        With a segmenter the mask stays on the GPU (statistics from mask_post, contour features from lmx_k_contour_features,
        128 bytes per frame come back); the rectangle fallback (no checkpoint / failing segmenter, main.py:90-100) is host code."""
        h, w = frames_host.shape[1:3]
        if self.sam_predictor is not None:
            try:
                dev = self.sam_predictor.device
                out = self.sam_predictor.segment_records(torch.from_numpy(frames_host).to(dev), torch.tensor(boxes, dtype=torch.float32, device=dev))
                stats, cont = out["stats"].cpu().tolist(), out["contour"].cpu().tolist()
                return [features_from_device(stats[i], cont[i], h, w) for i in range(len(boxes))]
            except Exception as e:  # noqa: BLE001 — main.py:90-92: a failing segmenter falls back to the rectangle
                print(f"SAM3 segmentation error: {e}")
        return [extract_segmentation_features(fallback_segmentation((h, w), b)) for b in boxes]

    @staticmethod
    def first_boxes(yolo_results):
        """{frame: bbox of the FIRST detection} (main.py:199-206; first = highest confidence, Appendix C-3)."""
        by_frame = {}
        if yolo_results and "detections" in yolo_results:
            for det in yolo_results["detections"]:
                if det["frame"] not in by_frame and det["detections"]:
                    by_frame[det["frame"]] = det["detections"][0]["bbox"]
        return by_frame

    @staticmethod
    def results_from_features(ids, fps, total, feats_by_frame):
        """main.py:208-254: per sampled frame a segmentation entry, the four means over the frames that have a mask."""
        segmentations, frame_features = [], []
        for i in ids:
            t = i / fps if fps > 0 else 0
            if i in feats_by_frame:
                feats = dict(feats_by_frame[i])
                feats["frame"] = i
                feats["time"] = t
                frame_features.append(feats)
                segmentations.append({"frame": i, "time": t, "mask_available": True, "features": feats})
            else:
                segmentations.append({"frame": i, "time": t, "mask_available": False})
        avg = {}
        if frame_features:
            avg = {"avg_mask_area": float(np.mean([f["mask_area"] for f in frame_features])),
                   "avg_area_ratio": float(np.mean([f["area_ratio"] for f in frame_features])),
                   "avg_circularity": float(np.mean([f["circularity"] for f in frame_features])),
                   "avg_aspect_ratio": float(np.mean([f["aspect_ratio"] for f in frame_features]))}
        return {"segmentations": segmentations, "aggregated_features": avg, "total_frames": total, "fps": fps,
                "frames_processed": len(segmentations)}

    async def write_and_publish(self, video_id, results):
        results_file = self.results_dir / f"{video_id}_sam3.json"
        with open(results_file, "w") as f:
            json.dump(results, f, indent=2)
        await self.nats_client.publish(self.config["nats"]["subjects"]["pipeline_sam3"], {
            "video_id": video_id, "pipeline": "sam3", "results_path": str(results_file), "features": results["aggregated_features"],
            "num_segmentations": len(results["segmentations"])})

    async def process_video(self, video_data):
        video_id = video_data["video_id"]
        processed_path = Path(video_data["processed_path"])
        if not processed_path.exists():
            print(f"Processed video not found: {processed_path}")
            return
        try:
            by_frame = self.first_boxes(await self.get_yolo_results(video_id))
            clip = R.Clip.open(processed_path)
            fps, total = clip.fps, clip.total_frames
            interval = max(1, fps // 2)
            ids, feats = [], {}
            pend_ids, pend_frames = [], []

            def flush():
                if pend_ids:
                    for i, f in zip(pend_ids, self._features(np.stack(pend_frames, 0), [by_frame[i] for i in pend_ids])):
                        feats[i] = f
                    pend_ids.clear()
                    pend_frames.clear()

            for i, frame in clip.iter_frames(lambda k: k % interval == 0):  # one pass; only frames with a box are kept
                ids.append(i)
                if by_frame.get(i):
                    pend_ids.append(i)
                    pend_frames.append(frame)
                    if len(pend_ids) == self.batch:
                        flush()
            flush()
            await self.write_and_publish(video_id, self.results_from_features(ids, fps, total, feats))
        except Exception as e:  # noqa: BLE001
            print(f"Error in SAM3 pipeline for {video_id}: {e}")
            traceback.print_exc()

    async def start(self):
        await self.nats_client.connect()
        await self.nats_client.subscribe(self.config["nats"]["subjects"]["video_preprocessed"], self.process_video)


class HieraSegmenter:
    """Adapter giving an image encoder (HieraEncoder, or SamVitEncoder for `sam_vit_b/l` checkpoints: main.py:58-65) +
    MaskDecoder the `segment(frames, boxes)` surface the service needs."""

    def __init__(self, encoder, decoder):
        self.encoder, self.decoder = encoder, decoder
        self.device = encoder.device

    def segment_records(self, frames, boxes):
        """-> dict(mask u8 [n,h,w], stats int64 [n,8], contour int64 [n,8], iou f32 [n]) on the device."""
        from .. import sam

        n, h, w, _ = frames.shape
        enc = self.encoder.encode(frames)
        e2 = enc["fpn"][2]
        d = self.decoder.predict(e2.view(-1, e2.shape[-1]), boxes, (h, w), sam.resize_longest_side(h, w, self.encoder.cfg.image))
        return dict(mask=d["mask"], stats=d["stats"], contour=K.contour_features(d["mask"]), iou=d["iou"])

    def segment(self, frames, boxes):
        return self.segment_records(frames, boxes)["mask"]


SamSegmenter = HieraSegmenter  # the adapter only relies on encode(...)["fpn"][2] being the [n,64,64,256] embedding
