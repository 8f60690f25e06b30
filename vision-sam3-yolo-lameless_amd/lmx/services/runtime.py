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
    (services/yolo-pipeline/app/main.py:55-71).  Sources: a video file via cv2 when it is installed, or an .npz written
    by ``save_npz_clip`` (frames uint8 [n,h,w,3] BGR, fps float) — the offline stand-in for decoded video."""

    def __init__(self, frames, fps):
        self.frames = frames
        self.fps = int(fps)  # the reference truncates: int(cap.get(CAP_PROP_FPS)) (Appendix C-1)
        self.total_frames = int(len(frames))

    _last = (None, None)  # (key, clip): the fused service opens the same clip three times in a row

    @staticmethod
    def open(path):
        path = str(path)
        key = (path, os.path.getmtime(path) if os.path.exists(path) else None)
        if Clip._last[0] == key:
            return Clip._last[1]
        c = Clip._open(path)
        Clip._last = (key, c)
        return c

    @staticmethod
    def _open(path):
        if path.endswith(".npz"):
            z = np.load(path)
            return Clip(z["frames"], float(z["fps"]))
        try:
            import cv2
        except ImportError as e:
            raise Exception(f"Failed to open video: {path} (cv2 is not installed and the file is not an .npz clip)") from e
        cap = cv2.VideoCapture(path)
        if not cap.isOpened():
            raise Exception(f"Failed to open video: {path}")
        fps = cap.get(cv2.CAP_PROP_FPS)
        frames = []
        while True:
            ok, f = cap.read()
            if not ok:
                break
            frames.append(f)
        total = int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
        cap.release()
        c = Clip(np.stack(frames, 0) if frames else np.zeros((0, 0, 0, 3), np.uint8), fps)
        c.total_frames = total
        return c


def save_npz_clip(path, frames, fps):
    np.savez(path, frames=np.asarray(frames, np.uint8), fps=np.float64(fps))


def sampled(n_decoded, interval):
    """Indices the reference's `frame_count % frame_interval == 0` loop visits."""
    return list(range(0, n_decoded, interval))


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
