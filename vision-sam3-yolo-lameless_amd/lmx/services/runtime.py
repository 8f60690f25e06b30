"""Plumbing shared by the service mirrors: configuration, message bus, clip reader, vector store.

External packages the reference imports (nats-py, cv2, qdrant_client) are absent offline; each has an
adapter here that uses the real package when importable and otherwise a small in-process equivalent with the same
observable contract, so that the service logic itself runs and is tested without them."""
import asyncio
import json
import os
from pathlib import Path

import numpy as np
import yaml

DEFAULT_SUBJECTS = {
    "video_preprocessed": "video.preprocessed",
    "pipeline_yolo": "pipeline.yolo",
    "pipeline_sam3": "pipeline.sam3",
    "pipeline_dinov3": "pipeline.dinov3",
}


def load_config(path="/app/shared/config/config.yaml"):
    """shared/config/config.yaml (reference keys: nats.subjects.*, models.yolo.confidence_threshold, qdrant.*).
    A missing file yields the reference's defaults (services/yolo-pipeline/app/main.py:44-49 returns {})."""
    p = Path(path)
    cfg = {}
    if p.exists():
        with open(p) as f:
            cfg = yaml.safe_load(f) or {}
    cfg.setdefault("nats", {})
    cfg["nats"].setdefault("url", os.getenv("NATS_URL", "nats://localhost:4222"))
    if os.getenv("NATS_URL"):
        cfg["nats"]["url"] = os.getenv("NATS_URL")
    subj = dict(DEFAULT_SUBJECTS)
    subj.update(cfg["nats"].get("subjects") or {})
    cfg["nats"]["subjects"] = subj
    return cfg


class InProcessBus:
    """Core-NATS semantics in one process: at-most-once, JSON payloads, handler exceptions swallowed
    (shared/utils/nats_client.py:47-70).  ``published`` records (subject, decoded payload) for inspection."""

    def __init__(self):
        self.handlers = {}
        self.published = []

    async def connect(self, url=None):
        return self

    async def publish(self, subject, data):
        wire = json.dumps(data).encode()  # the payload must be JSON-serialisable, exactly like the reference's publish
        self.published.append((subject, json.loads(wire.decode())))
        for cb in self.handlers.get(subject, []):
            try:
                await cb(json.loads(wire.decode()))
            except Exception as e:  # noqa: BLE001 — the reference's message_handler prints and drops
                print(f"Error processing message: {e}")

    async def subscribe(self, subject, callback):
        self.handlers.setdefault(subject, []).append(callback)

    async def close(self):
        self.handlers.clear()


class NATSClient:
    """Same surface as shared/utils/nats_client.py (connect / publish / subscribe / close) over nats-py."""

    def __init__(self, url):
        self.url = url
        self.nc = None

    async def connect(self, url=None):
        import nats  # raises ImportError offline: use InProcessBus there

        self.nc = await nats.connect(url or self.url)
        return self.nc

    async def publish(self, subject, data):
        if not self.nc:
            await self.connect()
        await self.nc.publish(subject, json.dumps(data).encode())

    async def subscribe(self, subject, callback):
        if not self.nc:
            await self.connect()

        async def handler(msg):
            try:
                await callback(json.loads(msg.data.decode()))
            except Exception as e:  # noqa: BLE001
                print(f"Error processing message: {e}")

        return await self.nc.subscribe(subject, cb=handler)

    async def close(self):
        if self.nc:
            await self.nc.close()
            self.nc = None


class Clip:
    """What the services take from cv2.VideoCapture: int(fps), int(frame count) and the decoded BGR frames in order
    (services/yolo-pipeline/app/main.py:55-71) — as a STREAM: like the reference's `cap.read()` loop the reader holds one
    decoded frame at a time and keeps only the frames a caller samples, so a long clip costs memory for the sampled frames
    only (a 5-minute 1080p clip is 56 GB decoded, 1.1 GB at the services' sampling).  Sources: a video file via cv2 when it
    is installed, or an .npz written by ``save_npz_clip`` (frames uint8 [n,h,w,3] BGR, fps float) — the offline stand-in
    for decoded video."""

    def __init__(self, path=None, frames=None, fps=None):
        self.path = str(path) if path is not None else None
        self._frames = frames          # in-memory source (npz stand-in or a caller's array)
        self.n_decoded = None          # known once a pass over the clip has finished
        if frames is not None:
            self.fps = int(fps)        # the reference truncates: int(cap.get(CAP_PROP_FPS)) (Appendix C-1)
            self.total_frames = int(len(frames))
            self.n_decoded = self.total_frames
            self.frame_hw = tuple(frames.shape[1:3]) if len(frames) else (0, 0)

    @staticmethod
    def open(path):
        path = str(path)
        if path.endswith(".npz"):
            z = np.load(path)
            return Clip(path, z["frames"], float(z["fps"]))
        try:
            import cv2
        except ImportError as e:
            raise Exception(f"Failed to open video: {path} (cv2 is not installed and the file is not an .npz clip)") from e
        cap = cv2.VideoCapture(path)
        if not cap.isOpened():
            raise Exception(f"Failed to open video: {path}")
        c = Clip(path)
        c.fps = int(cap.get(cv2.CAP_PROP_FPS))
        c.total_frames = int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
        c.frame_hw = (int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT)), int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)))
        cap.release()
        return c

    def iter_frames(self, keep=None):
        """Yield (index, BGR frame) in decode order for the indices `keep(index)` accepts (default: all).  One pass over the
        file; frames that are not kept are decoded (a codec has to) and dropped at once."""
        if self._frames is not None:
            for i in range(len(self._frames)):
                if keep is None or keep(i):
                    yield i, self._frames[i]
            return
        import cv2

        cap = cv2.VideoCapture(self.path)
        if not cap.isOpened():
            raise Exception(f"Failed to open video: {self.path}")
        i = 0
        try:
            while True:
                ok, f = cap.read()
                if not ok:
                    break
                if keep is None or keep(i):
                    yield i, f
                i += 1
        finally:
            cap.release()
        self.n_decoded = i

    def batches(self, intervals, batch):
        """Yield (indices, frames uint8 [k,h,w,3]) of at most `batch` frames whose index is a multiple of ANY of `intervals`
        — the union of the services' schedules (`frame_count % frame_interval == 0`, yolo main.py:74, dinov3 main.py:127)."""
        ivs = tuple(int(v) for v in (intervals if isinstance(intervals, (tuple, list)) else (intervals,)))
        ids, buf = [], []
        for i, f in self.iter_frames(lambda k: any(k % iv == 0 for iv in ivs)):
            ids.append(i)
            buf.append(f)
            if len(ids) == batch:
                yield ids, np.stack(buf, 0)
                ids, buf = [], []
        if ids:
            yield ids, np.stack(buf, 0)


def save_npz_clip(path, frames, fps):
    np.savez(path, frames=np.asarray(frames, np.uint8), fps=np.float64(fps))


def sampled(n_decoded, interval):
    """Indices the reference's `frame_count % frame_interval == 0` loop visits."""
    return list(range(0, n_decoded, interval))


class PinnedRing:
    """Pinned-host staging ring for the upload of decoded frames: chunk i+1 is copied to the device on a copy stream while
    chunk i computes (tools/pcie_probe.py measured 554 frames/s with the overlap vs 520 with blocking
    uploads).  `upload(frames_numpy)` returns a device tensor that is valid on the CURRENT stream once it has waited on the
    returned event; a slot is reused only after the event of its previous upload has completed."""

    def __init__(self, device, slots=3):
        import torch

        self.device = torch.device(device)
        self.slots = [None] * slots
        self.events = [None] * slots
        self.copy_stream = torch.cuda.Stream(self.device)
        self.k = 0

    def upload(self, frames):
        import torch

        j = self.k % len(self.slots)
        self.k += 1
        nbytes = frames.nbytes
        if self.events[j] is not None:
            self.events[j].synchronize()  # the copy out of this slot has finished
        if self.slots[j] is None or self.slots[j].numel() < nbytes:
            self.slots[j] = torch.empty((nbytes,), dtype=torch.uint8).pin_memory()
        host = self.slots[j][:nbytes].view(frames.shape)
        host.numpy()[...] = frames
        with torch.cuda.stream(self.copy_stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.events[j] = ev
        dev.record_stream(torch.cuda.current_stream(self.device))
        return dev, ev


class MemoryVectorStore:
    """The slice of Qdrant the dinov3 service uses (COSINE collection, search top-k, upsert by id;
    services/dinov3-pipeline/app/main.py:70-93,165-186,228-246), in memory."""

    def __init__(self):
        self.points = {}

    def ensure_collection(self, dim):
        self.dim = dim

    def search(self, vector, top_k=5):
        q = np.asarray(vector, np.float64)
        out = []
        for pid, (v, payload) in self.points.items():
            d = float(np.dot(q, v) / (np.linalg.norm(q) * np.linalg.norm(v) + 1e-30))
            out.append({"video_id": payload.get("video_id", "unknown"), "score": d, "label": payload.get("label", None),
                        "metadata": payload.get("metadata", {})})
        out.sort(key=lambda r: -r["score"])
        return out[:top_k]

    def upsert(self, pid, vector, payload):
        self.points[pid] = (np.asarray(vector, np.float64), dict(payload))


class QdrantStore:
    """Adapter over qdrant_client with the same three calls (used when the package and a server are available)."""

    def __init__(self, url, collection):
        from qdrant_client import QdrantClient  # ImportError offline

        self.client = QdrantClient(url=url)
        self.collection = collection

    def ensure_collection(self, dim):
        from qdrant_client.models import Distance, VectorParams

        names = [c.name for c in self.client.get_collections().collections]
        if self.collection not in names:
            self.client.create_collection(collection_name=self.collection, vectors_config=VectorParams(size=dim, distance=Distance.COSINE))

    def search(self, vector, top_k=5):
        res = self.client.search(collection_name=self.collection, query_vector=list(map(float, vector)), limit=top_k)
        return [{"video_id": r.payload.get("video_id", "unknown"), "score": float(r.score), "label": r.payload.get("label", None),
                 "metadata": r.payload.get("metadata", {})} for r in res]

    def upsert(self, pid, vector, payload):
        from qdrant_client.models import PointStruct

        self.client.upsert(collection_name=self.collection, points=[PointStruct(id=pid, vector=list(map(float, vector)), payload=payload)])
