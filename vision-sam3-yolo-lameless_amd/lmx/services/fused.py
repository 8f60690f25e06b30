"""FusedFeatureService — the three services as ONE subscriber of `video.preprocessed`: the clip is
opened and decoded ONCE (the reference decodes it three times: yolo main.py:55-71, sam3 main.py:180-194, dinov3
main.py:119-136), only the frames on the union of the services' schedules are kept, chunk i+1 is uploaded from a pinned
host ring on a copy stream while chunk i runs through `FusedExtractor.step`, and the three result files / subjects are
produced from that one pass in the order yolo -> sam3 -> dinov3 (ml-pipeline triggers on `pipeline.dinov3` and expects the
others on disk).  Running YOLO first also removes the reference's race between the yolo and sam3 services (Appendix C-4).

Multi-GPU: with torch.distributed initialised, every rank opens the clip and keeps the frames of its
contiguous block of the frame range; after the pass ONE collective gathers the fixed-stride per-frame records
(lmx.dist.gather_clip_records) to rank 0, which writes the three JSONs and publishes.  The other ranks publish nothing.

Schedules: "reference" runs YOLO + SAM on frames i % (fps//2) == 0 and DINO on i % fps == 0, like the services; "dense" runs
all three networks on every decoded frame (the throughput mode of BASELINE cfg#5) and derives the same JSONs from the
scheduled subset."""
import traceback
from pathlib import Path

import numpy as np
import torch
import torch.distributed as tdist

from .. import dist as ldist
from . import runtime as R
from .sam3_pipeline import extract_segmentation_features, features_from_device
from .yolo_pipeline import detections_from_device


class FusedFeatureService:
    def __init__(self, extractor, yolo_pipeline, sam3_pipeline, dinov3_pipeline, schedule="reference", chunk=32):
        """extractor: object with .device and .step(frames, conf=, det_idx=, emb_idx=) -> dict of per-frame device tensors
        (lmx.pipeline.FusedExtractor).  The three pipeline objects supply configuration, result directories, the JSON
        builders and the publishers; their own model handles are not used."""
        if schedule not in ("reference", "dense"):
            raise ValueError(f"schedule {schedule!r}: expected 'reference' or 'dense'")
        self.fx = extractor
        self.yolo, self.sam3, self.dinov3 = yolo_pipeline, sam3_pipeline, dinov3_pipeline
        self.nats_client = yolo_pipeline.nats_client
        self.config = yolo_pipeline.config
        self.schedule, self.chunk = schedule, chunk
        self._ring = None

    # ---- one pass over the clip on this rank ---------------------------------------------------------------------------
    def _upload(self, host):
        dev = torch.device(self.fx.device)
        if dev.type != "cuda":  # CPU stand-in extractors of the host tests
            return torch.from_numpy(host), None
        if self._ring is None:
            self._ring = R.PinnedRing(dev)
        return self._ring.upload(host)

    def _run_clip(self, clip, rank, world):
        fps, total = clip.fps, clip.total_frames
        i_det, i_emb = max(1, fps // 2), max(1, fps)
        per = -(-max(total, 1) // world)
        mine = (lambda i: min(i // per, world - 1) == rank)
        on_sched = (lambda i: i % i_det == 0 or i % i_emb == 0)
        keep = (lambda i: mine(i) and (self.schedule == "dense" or on_sched(i)))
        conf = self.yolo.confidence_threshold
        parts, pending = [], None

        def launch(ids, dev_frames, ev):
            if ev is not None:
                torch.cuda.current_stream(dev_frames.device).wait_event(ev)
            if self.schedule == "dense":
                out = dict(self.fx.step(dev_frames, conf=conf))
                n = len(ids)
                out["ran_det"] = torch.ones((n,), dtype=torch.int32, device=dev_frames.device)
                out["ran_emb"] = torch.ones((n,), dtype=torch.int32, device=dev_frames.device)
            else:
                det = [j for j, i in enumerate(ids) if i % i_det == 0]
                emb = [j for j, i in enumerate(ids) if i % i_emb == 0]
                out = dict(self.fx.step(dev_frames, conf=conf, det_idx=det, emb_idx=emb))
            out.pop("mask", None)
            out["frame_id"] = torch.as_tensor(ids, dtype=torch.int64, device=dev_frames.device)
            parts.append(out)

        # software pipeline: the upload of chunk i+1 is issued (copy stream) before chunk i is launched
        for ids, host in self._kept_batches(clip, keep):
            up = self._upload(host)
            if pending is not None:
                launch(*pending)
            pending = (ids, up[0], up[1])
        if pending is not None:
            launch(*pending)
        return parts

    def _kept_batches(self, clip, keep):
        ids, buf = [], []
        for i, f in clip.iter_frames(keep):
            ids.append(i)
            buf.append(f)
            if len(ids) == self.chunk:
                yield ids, np.stack(buf, 0)
                ids, buf = [], []
        if ids:
            yield ids, np.stack(buf, 0)

    @staticmethod
    def _concat(parts, template):
        if parts:
            return {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
        return {k: v[:0] for k, v in template.items()}

    def _empty_template(self, clip):
        """Zero-row record with the step's field shapes (a rank whose block holds no scheduled frame still joins the gather)."""
        dev = torch.device(self.fx.device)
        h, w = clip.frame_hw
        z = torch.zeros
        D = getattr(getattr(self.fx, "dino", None), "cfg", None)
        D = D.hidden if D is not None else self.dinov3.model.cfg.hidden
        return dict(boxes=z((0, 300, 4), device=dev), scores=z((0, 300), device=dev), cls=z((0, 300), dtype=torch.int32, device=dev),
                    counts=z((0,), dtype=torch.int32, device=dev), embedding=z((0, D), device=dev),
                    mask_bits=z((0, h, (w + 7) // 8), dtype=torch.uint8, device=dev), mask_stats=z((0, 8), dtype=torch.int64, device=dev),
                    mask_contour=z((0, 8), dtype=torch.int64, device=dev),
                    mask_iou=z((0,), device=dev), ran_det=z((0,), dtype=torch.int32, device=dev),
                    ran_emb=z((0,), dtype=torch.int32, device=dev), frame_id=z((0,), dtype=torch.int64, device=dev))

    # ---- the three result files from the gathered records (rank 0) ---------------------------------------------------------
    async def _emit(self, video_data, clip, rec):
        video_id = video_data["video_id"]
        fps, total = clip.fps, clip.total_frames
        i_det, i_emb = max(1, fps // 2), max(1, fps)
        host = {k: v.cpu() for k, v in rec.items()}  # one D2H per field for the whole clip
        fid = host["frame_id"].tolist()
        order = sorted(range(len(fid)), key=lambda j: fid[j])
        det_rows = [j for j in order if fid[j] % i_det == 0 and int(host["ran_det"][j])]
        emb_rows = [j for j in order if fid[j] % i_emb == 0 and int(host["ran_emb"][j])]
        names = getattr(self.yolo.yolo_model, "names", None) or getattr(getattr(self.fx, "yolo", None), "names", {})
        # yolo
        sel = torch.as_tensor(det_rows, dtype=torch.int64)
        dets = detections_from_device([fid[j] for j in det_rows], fps, names, host["boxes"][sel], host["scores"][sel],
                                      host["cls"][sel], host["counts"][sel]) if det_rows else []
        yolo_results = self.yolo.results_from_detections(dets, total, fps)
        await self.yolo.write_and_publish(video_id, yolo_results)
        self.sam3.yolo_results_cache[video_id] = yolo_results
        # sam3: a mask exists where YOLO found something (the prompt is its first box).  The 7 features come from what the
        # device computed (mask_post's statistics + lmx_k_contour_features): the mask itself never leaves the GPU; an
        # extractor without the contour record falls back to the host border following on the unpacked bits
        h, w = clip.frame_hw
        feats = {}
        for j in det_rows:
            if int(host["counts"][j]) > 0:
                if "mask_contour" in host:
                    feats[fid[j]] = features_from_device(host["mask_stats"][j].tolist(), host["mask_contour"][j].tolist(), h, w)
                else:
                    m = np.unpackbits(host["mask_bits"][j].numpy(), axis=-1, count=w).astype(bool)
                    feats[fid[j]] = extract_segmentation_features(m)
        await self.sam3.write_and_publish(video_id, self.sam3.results_from_features([fid[j] for j in det_rows], fps, total, feats))
        # dinov3
        embs = [{"frame": fid[j], "time": fid[j] / fps if fps > 0 else 0, "embedding": host["embedding"][j].numpy().tolist()}
                for j in emb_rows]
        await self.dinov3.finish(video_data, self.dinov3.embeddings_result(embs, total, fps))

    async def process_video(self, video_data):
        video_id = video_data["video_id"]
        processed_path = Path(video_data["processed_path"])
        multi = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
        rank, world = (tdist.get_rank(), tdist.get_world_size()) if multi else (0, 1)
        # The local pass has its own try block: whatever happens on this rank (missing file, decode error, OOM, LmxError), it
        # still reaches the control all-reduce below, which carries an error flag beside the row count.  A rank that returned
        # early would leave its peers blocked in the collective forever.
        clip = rec = None
        failed = 0
        try:
            if not processed_path.exists():
                print(f"Processed video not found: {processed_path}")
                failed = 1
            else:
                clip = R.Clip.open(processed_path)          # ONE open, ONE decode pass per rank
                parts = self._run_clip(clip, rank, world)
                rec = self._concat(parts, self._empty_template(clip))
        except Exception as e:  # noqa: BLE001 — like the services, never raise out of the handler
            print(f"Error in fused pipeline for {video_id}: {e}")
            traceback.print_exc()
            failed = 1
        try:
            if multi:
                # every rank must contribute equally many rows: agree on the largest shard AND on whether anybody failed
                # (16 bytes of control traffic, one all-reduce), then ONE gather of the packed records — or none at all,
                # on every rank together, when a rank failed (then no file is written and nothing is published, as in the
                # single-process case)
                ctl = torch.tensor([0 if rec is None else int(rec["frame_id"].shape[0]), failed], dtype=torch.int64)
                if tdist.get_backend() != "gloo":
                    ctl = ctl.to(torch.device(self.fx.device))
                tdist.all_reduce(ctl, op=tdist.ReduceOp.MAX)
                n_rows, any_failed = (int(v) for v in ctl.tolist())
                if any_failed:
                    if not failed:
                        print(f"Fused pipeline for {video_id}: another rank failed; nothing is written")
                    return
                buf, layout = ldist.pack_records(rec, n_rows)
                g = ldist.gather_packed(buf, root=0)
                if g is None:
                    return
                rec = ldist.unpack_records(g, layout)
            elif failed:
                return
            await self._emit(video_data, clip, rec)
        except Exception as e:  # noqa: BLE001
            print(f"Error in fused pipeline for {video_id}: {e}")
            traceback.print_exc()

    async def start(self):
        await self.nats_client.connect()
        await self.nats_client.subscribe(self.config["nats"]["subjects"]["video_preprocessed"], self.process_video)
