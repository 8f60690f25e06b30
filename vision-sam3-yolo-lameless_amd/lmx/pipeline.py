"""The fused per-frame path (BASELINE cfg#5): decoded 1080p BGR frames, resident in HBM, go through the three networks
of services/{yolo,sam3,dinov3}-pipeline on one GPU.  Frames are independent, so a clip is sharded
across ranks as contiguous blocks and each rank runs this object on its block; lmx.dist reassembles per-clip records."""
import os

import torch

from . import dino, sam, sam_decoder, yolo
from . import kernels as K

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")  # package data: BatchNorm statistics of the synthetic YOLO weights


def stream_plan(n_chunks, max_streams=4, layout="lanes"):
    """-> (pool size, YOLO stream, DINO stream, [stream of each SAM pass]) as indices into the stream pool.
    "lanes": YOLO and DINO on a stream each, the SAM passes dealt over the remaining max_streams - 2 streams (two passes
    in flight at the default of 4); "rr": everything dealt round-robin over max_streams streams."""
    if layout not in ("lanes", "rr"):
        raise ValueError(f"stream layout {layout!r}: expected 'lanes' or 'rr'")
    k = max(3, min(2 + n_chunks, int(max_streams)))
    if layout == "rr":
        return k, 0, 1, [(2 + j) % k for j in range(n_chunks)]
    return k, 0, 1, [2 + j % (k - 2) for j in range(n_chunks)]


class FusedExtractor:
    def __init__(self, device="cuda", yolo_scale="l", weight_seeds=(7, 5, 3), yolo_bn=None):
        """Synthetic weights (no checkpoints exist offline): YOLOv8-{scale}, Hiera-B+, DINOv3 ViT-L/16."""
        from . import weights

        self.device = torch.device(device)
        self.serial = bool(os.environ.get("LMX_SERIAL"))  # True: every launch on the caller's stream
        # HIP streams one step may keep in flight (>= 3): YOLO, DINO and up to four SAM passes.  Round 1 capped this at 4 because
        # more SAM passes in flight showed wrong mask pixels; round 2 traced that to an instruction form the build now forbids
        # (DESIGN.md section 6), so the cap is a performance setting again: 6 measures +3.8 % over 4; 7 = YOLO, DINO and the
        # five SAM passes of 30 frames of a 150-frame clip.
        self.max_streams = int(os.environ.get("LMX_MAX_STREAMS", "7"))
        self.stream_layout = os.environ.get("LMX_STREAM_LAYOUT", "lanes")
        ycfg = yolo.YoloConfig(yolo_scale)
        bn = yolo_bn or os.path.join(DATA, f"yolov8{yolo_scale}_bn_w{weight_seeds[0]}.npz")
        self.yolo = yolo.YoloDetector(ycfg, yolo.synthetic_state_dict(ycfg, weight_seeds[0], bn if os.path.exists(bn) else None),
                                      self.device)
        scfg = sam.hiera_b_plus()
        self.sam = sam.HieraEncoder(scfg, weights.synth_state_dict(sam.param_spec(scfg), weight_seeds[1]), self.device)
        self.decoder = sam_decoder.MaskDecoder(sam_decoder.synthetic_state_dict(weight_seeds[1] + 100), self.device)
        dcfg = dino.dinov3_vitl16()
        self.dino = dino.DinoEmbedder(dcfg, weights.synth_state_dict(dino.param_spec(dcfg), weight_seeds[2]), self.device)

    @classmethod
    def from_models(cls, detector, sam_encoder, mask_decoder, embedder):
        """The fused path over models that are already on the device (any YOLOv8 scale, Hiera or SAM-ViT encoder, DINOv2/3)."""
        self = cls.__new__(cls)
        self.device = detector.device
        self.serial = bool(os.environ.get("LMX_SERIAL"))
        self.max_streams = int(os.environ.get("LMX_MAX_STREAMS", "7"))
        self.stream_layout = os.environ.get("LMX_STREAM_LAYOUT", "lanes")
        self.yolo, self.sam, self.decoder, self.dino = detector, sam_encoder, mask_decoder, embedder
        return self

    def _streams(self, k):
        pool = getattr(self, "_pool", None)
        if pool is None:
            pool = self._pool = []
        while len(pool) < k:
            pool.append(torch.cuda.Stream(self.device))
        return pool[:k]

    def step(self, frames, conf=0.5, sam_chunk=16, keep_byte_masks=False, det_idx=None, emb_idx=None, precision=None):
        """frames u8 [n,1080,1920,3] BGR on device -> dict of device tensors for every frame.  Masks are returned bit-packed
        ([n, h, ceil(w/8)], numpy.packbits order): that is what is gathered across GPUs and copied to the host;
        `keep_byte_masks` adds the u8 [n,h,w] masks the kernels produced.
        Dense schedule (default): every frame goes through all three networks.  Reference schedule: `det_idx` / `emb_idx`
        (index lists into `frames`) name the frames YOLO + SAM resp. DINO run on (yolo main.py:74, dinov3 main.py:127); the
        outputs keep n rows, zero where a network did not run, plus `ran_det` / `ran_emb` flags.
        precision: None = the models' default plans ("exact": YOLO keep-sets of the fp32 path and masks within IoU 0.9995 of it —
        what the services' JSON must carry) or "f16" (the throughput plans; bench.py's dense step) — lmx.yolo.YoloDetector,
        lmx.sam_decoder.MaskDecoder."""
        n = frames.shape[0]
        if det_idx is None and emb_idx is None:
            return self._step_dense(frames, conf, sam_chunk, keep_byte_masks, precision=precision)
        dev = frames.device
        di = torch.as_tensor(list(range(n)) if det_idx is None else list(det_idx), dtype=torch.int64, device=dev)
        ei = torch.as_tensor(list(range(n)) if emb_idx is None else list(emb_idx), dtype=torch.int64, device=dev)
        h, w = frames.shape[1:3]
        out = dict(boxes=torch.zeros((n, 300, 4), dtype=torch.float32, device=dev), scores=torch.zeros((n, 300), dtype=torch.float32, device=dev),
                   cls=torch.zeros((n, 300), dtype=torch.int32, device=dev), counts=torch.zeros((n,), dtype=torch.int32, device=dev),
                   embedding=torch.zeros((n, self.dino.cfg.hidden), dtype=torch.float32, device=dev),
                   mask_bits=torch.zeros((n, h, (w + 7) // 8), dtype=torch.uint8, device=dev),
                   mask_stats=torch.zeros((n, 8), dtype=torch.int64, device=dev), mask_contour=torch.zeros((n, 8), dtype=torch.int64, device=dev),
                   mask_iou=torch.zeros((n,), dtype=torch.float32, device=dev),
                   ran_det=torch.zeros((n,), dtype=torch.int32, device=dev), ran_emb=torch.zeros((n,), dtype=torch.int32, device=dev))
        fd = frames if det_idx is None else frames.index_select(0, di)
        fe = frames if emb_idx is None else frames.index_select(0, ei)
        part = self._step_dense(fd, conf, sam_chunk, keep_byte_masks, emb_frames=fe, precision=precision)
        if di.numel():
            for k in ("boxes", "scores", "cls", "counts", "mask_bits", "mask_stats", "mask_contour", "mask_iou"):
                out[k].index_copy_(0, di, part[k].to(out[k].dtype))
            out["ran_det"].index_fill_(0, di, 1)
            if keep_byte_masks:
                out["mask"] = torch.zeros((n, h, w), dtype=torch.uint8, device=dev)
                out["mask"].index_copy_(0, di, part["mask"])
        if ei.numel():
            out["embedding"].index_copy_(0, ei, part["embedding"])
            out["ran_emb"].index_fill_(0, ei, 1)
        return out

    def _step_dense(self, frames, conf, sam_chunk, keep_byte_masks, emb_frames=None, precision=None):
        """YOLO -> top-1 box -> SAM on every frame of `frames`; DINO on every frame of `emb_frames` (default: the same)."""
        n, h, w, _ = frames.shape
        emb_frames = frames if emb_frames is None else emb_frames
        rhw = sam.resize_longest_side(h, w, self.sam.cfg.image)
        # The three networks only meet at the mask decoder (SAM's prompt = YOLO's top box), and frames are independent.
        # YOLO, DINO and every SAM chunk of `sam_chunk` frames each run on a HIP stream of their own: MFMA-bound GEMMs of one
        # stream fill the CUs while another stream is in HBM-bound kernels (LayerNorm, narrow-stage attention, residual
        # epilogues), and the tail of one launch overlaps the head of another.  `self.serial` (LMX_SERIAL=1) keeps one stream.
        main = torch.cuda.current_stream(self.device)
        chunks = list(range(0, n, sam_chunk))
        if self.serial:
            det_stream, emb_stream, sam_streams = main, main, [main] * len(chunks)
        else:
            n_pool, di, ei, si = stream_plan(max(1, len(chunks)), self.max_streams, self.stream_layout)
            pool = self._streams(n_pool)
            det_stream, emb_stream, sam_streams = pool[di], pool[ei], [pool[j] for j in si][:len(chunks)]
            for st in pool:
                st.wait_stream(main)  # frames were produced on the caller's stream
        empty = n == 0
        with torch.cuda.stream(det_stream):
            if empty:
                z = torch.zeros
                boxes, scores, cls, counts = (z((0, 300, 4), device=self.device), z((0, 300), device=self.device),
                                              z((0, 300), dtype=torch.int32, device=self.device), z((0,), dtype=torch.int32, device=self.device))
            else:
                boxes, scores, cls, src, counts = self.yolo.detect(frames, conf=conf, precision=precision)
            det_done = torch.cuda.Event()
            det_done.record(det_stream)
        with torch.cuda.stream(emb_stream):
            emb = (self.dino.embed_frames(emb_frames) if emb_frames.shape[0]
                   else torch.zeros((0, self.dino.cfg.hidden), dtype=torch.float32, device=self.device))
        masks, stats, ious, conts = [], [], [], []
        for i, st in zip(chunks, sam_streams):  # Hiera activations are ~100 MB/frame: a chunk bounds the live set
            with torch.cuda.stream(st):
                enc = self.sam.encode(frames[i:i + sam_chunk], precision=precision)
                e2 = enc["fpn"][2]
                st.wait_event(det_done)  # the decoder needs the boxes
                # the service prompts SAM with the first (highest-confidence) detection of the frame (sam3 main.py:199-206);
                # frames without a detection are decoded against an all-zero box and flagged by counts == 0
                d = self.decoder.predict(e2.view(-1, e2.shape[-1]), boxes[i:i + sam_chunk, 0, :], (h, w), rhw, precision=precision)
                # the contour part of extract_segmentation_features (sam3 main.py:118-135) on the device, on the pass's stream
                conts.append(K.contour_features(d["mask"]))
            masks.append(d["mask"])
            stats.append(d["stats"])
            ious.append(d["iou"])
        for st in set(sam_streams + [det_stream, emb_stream]):
            if st is not main:
                main.wait_stream(st)
        if empty:
            mask = torch.zeros((0, h, w), dtype=torch.uint8, device=self.device)
            out = dict(boxes=boxes, scores=scores, cls=cls, counts=counts, embedding=emb,
                       mask_bits=torch.zeros((0, h, (w + 7) // 8), dtype=torch.uint8, device=self.device),
                       mask_stats=torch.zeros((0, 8), dtype=torch.int64, device=self.device),
                       mask_contour=torch.zeros((0, 8), dtype=torch.int64, device=self.device),
                       mask_iou=torch.zeros((0,), dtype=torch.float32, device=self.device))
        else:
            cat = (lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, 0))
            mask = cat(masks)
            out = dict(boxes=boxes, scores=scores, cls=cls, counts=counts, embedding=emb, mask_bits=K.pack_bits(mask),
                       mask_stats=cat(stats), mask_contour=cat(conts), mask_iou=cat(ious))
        if keep_byte_masks:
            out["mask"] = mask
        return out
