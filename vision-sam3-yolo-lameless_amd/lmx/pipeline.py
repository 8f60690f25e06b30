"""The fused per-frame path (BASELINE cfg#5): decoded 1080p BGR frames, resident in HBM, go through the three networks
of services/{yolo,sam3,dinov3}-pipeline on one GPU.  Frames are independent, so a clip is sharded
across ranks as contiguous blocks and each rank runs this object on its block; lmx.dist reassembles per-clip records."""
import os

import torch

from . import dino, sam, sam_decoder, yolo
from . import kernels as K

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


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
        self.max_streams = int(os.environ.get("LMX_MAX_STREAMS", "4"))  # HIP streams one step may keep in flight (>= 3)
        self.stream_layout = os.environ.get("LMX_STREAM_LAYOUT", "lanes")
        ycfg = yolo.YoloConfig(yolo_scale)
        bn = yolo_bn or os.path.join(GOLDEN, f"yolov8{yolo_scale}_bn_w{weight_seeds[0]}.npz")
        self.yolo = yolo.YoloDetector(ycfg, yolo.synthetic_state_dict(ycfg, weight_seeds[0], bn if os.path.exists(bn) else None),
                                      self.device)
        scfg = sam.hiera_b_plus()
        self.sam = sam.HieraEncoder(scfg, weights.synth_state_dict(sam.param_spec(scfg), weight_seeds[1]), self.device)
        self.decoder = sam_decoder.MaskDecoder(sam_decoder.synthetic_state_dict(weight_seeds[1] + 100), self.device)
        dcfg = dino.dinov3_vitl16()
        self.dino = dino.DinoEmbedder(dcfg, weights.synth_state_dict(dino.param_spec(dcfg), weight_seeds[2]), self.device)

    def _streams(self, k):
        pool = getattr(self, "_pool", None)
        if pool is None:
            pool = self._pool = []
        while len(pool) < k:
            pool.append(torch.cuda.Stream(self.device))
        return pool[:k]

    def step(self, frames, conf=0.5, sam_chunk=16, keep_byte_masks=False):
        """frames u8 [n,1080,1920,3] BGR on device -> dict of device tensors for every frame (dense schedule).  Masks are
        returned bit-packed ([n, h, ceil(w/8)], numpy.packbits order): that is what is gathered across GPUs and copied to
        the host; `keep_byte_masks` adds the u8 [n,h,w] masks the kernels produced."""
        n, h, w, _ = frames.shape
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
            n_pool, di, ei, si = stream_plan(len(chunks), self.max_streams, self.stream_layout)
            pool = self._streams(n_pool)
            det_stream, emb_stream, sam_streams = pool[di], pool[ei], [pool[j] for j in si]
            for st in pool:
                st.wait_stream(main)  # frames were produced on the caller's stream
        with torch.cuda.stream(det_stream):
            boxes, scores, cls, src, counts = self.yolo.detect(frames, conf=conf)
            det_done = torch.cuda.Event()
            det_done.record(det_stream)
        with torch.cuda.stream(emb_stream):
            emb = self.dino.embed_frames(frames)
        masks, stats, ious = [], [], []
        for i, st in zip(chunks, sam_streams):  # Hiera activations are ~100 MB/frame: a chunk bounds the live set
            with torch.cuda.stream(st):
                enc = self.sam.encode(frames[i:i + sam_chunk])
                e2 = enc["fpn"][2]
                st.wait_event(det_done)  # the decoder needs the boxes
                # the service prompts SAM with the first (highest-confidence) detection of the frame (sam3 main.py:199-206);
                # frames without a detection are decoded against an all-zero box and flagged by counts == 0
                d = self.decoder.predict(e2.view(-1, e2.shape[-1]), boxes[i:i + sam_chunk, 0, :], (h, w), rhw)
            masks.append(d["mask"])
            stats.append(d["stats"])
            ious.append(d["iou"])
        for st in set(sam_streams + [det_stream, emb_stream]):
            if st is not main:
                main.wait_stream(st)
        cat = (lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, 0))
        mask = cat(masks)
        out = dict(boxes=boxes, scores=scores, cls=cls, counts=counts, embedding=emb, mask_bits=K.pack_bits(mask),
                   mask_stats=cat(stats), mask_iou=cat(ious))
        if keep_byte_masks:
            out["mask"] = mask
        return out
