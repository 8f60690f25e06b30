"""The tracking service on the gathered records: ByteTrack association over the YOLO JSON and Re-ID
over the DINO embedding — the consumer directly after the hot path (services/tracking-service/app/main.py:114-234,268-382,
app/tracker/{bytetrack,kalman,matching,track}.py, app/reid/matcher.py:104-150).

Same subjects (`pipeline.yolo`, `pipeline.dinov3` in; `tracking.complete`, `tracking.reid.match` out), same
`{video_id}_tracking.json`, same thresholds and the same life-cycle quirks (Appendix C style: they are the contract):
  * LOST tracks take part in the first association like active ones (`get_all_tracks`, bytetrack.py:104);
  * every track left unmatched by the FIRST stage is marked missed at the end of the frame even if the second stage matched
    it (bytetrack.py:147-149) — only a reactivated LOST track escapes;
  * `avg_confidence` is the track's last confidence (boxes have four numbers, main.py:196-197).
The per-frame arithmetic (IoU matrix, minimum-cost assignment) is the C-ABI's host code (csrc/host_track.cpp): tens of boxes,
sequential in the frame index — there is nothing for the GPU in it.  The Kalman filter is filterpy.kalman.KalmanFilter's
predict / update (Joseph-form covariance) written out for the 7-state constant-velocity box model of kalman.py:22-74.
filterpy, lap, qdrant_client and sqlalchemy are absent here: PARITY UNPINNED against them; the relational side
(`_save_track_to_db`, main.py:384-431) is the reference's storage layer and out of scope."""
import ctypes as C
import json
import os
import uuid

import numpy as np

from .. import _lib

TENTATIVE, CONFIRMED, LOST, DELETED = "TENTATIVE", "CONFIRMED", "LOST", "DELETED"


def iou_batch(a, b):
    """matching.py:12-44 through lmx_h_iou_matrix."""
    a = np.ascontiguousarray(np.atleast_2d(np.asarray(a, np.float64)))
    b = np.ascontiguousarray(np.atleast_2d(np.asarray(b, np.float64)))
    out = np.empty((a.shape[0], b.shape[0]), np.float64)
    rc = _lib.load().lmx_h_iou_matrix(a.ctypes.data_as(C.c_void_p), a.shape[0], b.ctypes.data_as(C.c_void_p), b.shape[0],
                                      out.ctypes.data_as(C.c_void_p))
    if rc:
        raise _lib.LmxError("lmx_h_iou_matrix: invalid argument")
    return out


def cosine_distance(f1, f2):
    """matching.py:47-66."""
    f1 = f1 / (np.linalg.norm(f1, axis=1, keepdims=True) + 1e-6)
    f2 = f2 / (np.linalg.norm(f2, axis=1, keepdims=True) + 1e-6)
    return 1.0 - f1 @ f2.T


def linear_assignment(cost):
    """matching.py:69-101: (matched [k,2], unmatched rows, unmatched cols) through lmx_h_assign."""
    cost = np.ascontiguousarray(cost, np.float64)
    n, m = cost.shape
    if cost.size == 0:
        return np.empty((0, 2), dtype=int), np.arange(n), np.arange(m)
    x, y = np.empty(n, np.int32), np.empty(m, np.int32)
    rc = _lib.load().lmx_h_assign(cost.ctypes.data_as(C.c_void_p), n, m, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    if rc:
        raise _lib.LmxError("lmx_h_assign: invalid argument (non-finite cost?)")
    matched = np.array([[i, j] for i, j in enumerate(x) if j >= 0], dtype=int).reshape(-1, 2)
    return matched, np.array([i for i, j in enumerate(x) if j < 0], dtype=int), np.array([j for j, i in enumerate(y) if i < 0], dtype=int)


def associate_detections_to_tracks(dets, tracks, iou_threshold=0.3, det_features=None, track_features=None, appearance_weight=0.5):
    """matching.py:104-173: minimum-cost assignment on 1 - IoU (optionally blended with the cosine distance of the
    appearance features), then matches under the IoU threshold are returned to the unmatched lists."""
    if len(tracks) == 0:
        return np.empty((0, 2), dtype=int), np.arange(len(dets)), np.empty(0, dtype=int)
    if len(dets) == 0:
        return np.empty((0, 2), dtype=int), np.empty(0, dtype=int), np.arange(len(tracks))
    iou = iou_batch(dets, tracks)
    if det_features is not None and track_features is not None:
        cost = (1 - appearance_weight) * (1.0 - iou) + appearance_weight * cosine_distance(det_features, track_features)
    else:
        cost = 1.0 - iou
    matched, un_d, un_t = linear_assignment(cost)
    valid = []
    for d, t in matched:
        if iou[int(d), int(t)] >= iou_threshold:
            valid.append([int(d), int(t)])
        else:
            un_d = np.append(un_d, int(d))
            un_t = np.append(un_t, int(t))
    return np.array(valid, dtype=int).reshape(-1, 2), un_d.astype(int), un_t.astype(int)


class KalmanBoxTracker:
    """kalman.py:15-141: state [x, y, s, r, vx, vy, vs] (centre, area, aspect ratio and the velocities of the first three),
    observation [x, y, s, r]; filterpy's defaults (x = 0, P = Q = R = I) with the scalings of kalman.py:52-63."""

    F = np.eye(7)
    F[0, 4] = F[1, 5] = F[2, 6] = 1.0
    H = np.eye(4, 7)

    def __init__(self, bbox):
        self.x = np.zeros((7, 1))
        self.P = np.eye(7)
        self.Q = np.eye(7)
        self.R = np.eye(4)
        self.R[2:, 2:] *= 10.0
        self.P[4:, 4:] *= 1000.0
        self.P *= 10.0
        self.Q[-1, -1] *= 0.01
        self.Q[4:, 4:] *= 0.01
        self.x[:4] = self._bbox_to_z(bbox)
        self.time_since_update = 0
        self.history = []
        self.hits = 0
        self.hit_streak = 0
        self.age = 0
        self.last_detection = bbox

    @staticmethod
    def _bbox_to_z(bbox):
        w = bbox[2] - bbox[0]
        h = bbox[3] - bbox[1]
        return np.array([[bbox[0] + w / 2], [bbox[1] + h / 2], [w * h], [w / (h + 1e-6)]])

    @staticmethod
    def _z_to_bbox(z):
        x, y, s, r = z.flatten()[:4]
        s = max(1e-6, s)
        r = max(1e-6, r)
        w = np.sqrt(s * r)
        h = s / (w + 1e-6)
        return np.array([x - w / 2, y - h / 2, x + w / 2, y + h / 2])

    def update(self, bbox):
        self.time_since_update = 0
        self.history = []
        self.hits += 1
        self.hit_streak += 1
        z = self._bbox_to_z(bbox)
        H, P, R = self.H, self.P, self.R
        y = z - H @ self.x
        PHT = P @ H.T
        S = H @ PHT + R
        K = PHT @ np.linalg.inv(S)
        self.x = self.x + K @ y
        I_KH = np.eye(7) - K @ H
        self.P = I_KH @ P @ I_KH.T + K @ R @ K.T
        self.last_detection = bbox

    def predict(self):
        if self.x[6] + self.x[2] <= 0:  # kalman.py:124-125: never predict a negative area
            self.x[6] = 0
        self.x = self.F @ self.x
        self.P = self.F @ self.P @ self.F.T + self.Q
        self.age += 1
        if self.time_since_update > 0:
            self.hit_streak = 0
        self.time_since_update += 1
        self.history.append(self._z_to_bbox(self.x))
        return self.history[-1]

    def get_state(self):
        return self._z_to_bbox(self.x)


class Track:
    """track.py:21-156."""

    def __init__(self, track_id, bbox, confidence=0.0, embedding=None, frame_idx=0):
        self.track_id = track_id
        self.bbox = bbox
        self.confidence = confidence
        self.embedding = embedding
        self.state = TENTATIVE
        self.age = 0
        self.hits = 1
        self.time_since_update = 0
        self.frame_history = [frame_idx]
        self.bbox_history = [bbox.copy()]
        self.smoothed_embedding = embedding.copy() if embedding is not None else None

    def update(self, bbox, confidence, embedding=None, frame_idx=0):
        self.bbox = bbox.copy()
        self.confidence = confidence
        self.hits += 1
        self.time_since_update = 0
        self.bbox_history.append(bbox.copy())
        self.frame_history.append(frame_idx)
        if embedding is not None:
            if self.smoothed_embedding is None:
                self.smoothed_embedding = embedding.copy()
            else:
                self.smoothed_embedding = 0.9 * self.smoothed_embedding + (1 - 0.9) * embedding
            self.embedding = embedding
        if self.state == TENTATIVE and self.hits >= 3:
            self.state = CONFIRMED
        elif self.state == LOST:
            self.state = CONFIRMED

    def mark_missed(self):
        self.age += 1
        self.time_since_update += 1
        if self.state == CONFIRMED and self.time_since_update > 30:
            self.state = LOST
        elif self.state == TENTATIVE and self.time_since_update > 3:
            self.state = DELETED
        elif self.state == LOST and self.time_since_update > 90:
            self.state = DELETED

    def predict(self, predicted_bbox):
        self.bbox = predicted_bbox
        self.age += 1

    def get_feature(self):
        return self.smoothed_embedding

    def to_dict(self):
        return {"track_id": self.track_id, "bbox": self.bbox.tolist(), "confidence": float(self.confidence), "state": self.state,
                "age": self.age, "hits": self.hits, "time_since_update": self.time_since_update,
                "start_frame": self.frame_history[0] if self.frame_history else 0,
                "end_frame": self.frame_history[-1] if self.frame_history else 0, "has_embedding": self.embedding is not None}


class Detection:
    """bytetrack.py:26-32."""

    def __init__(self, bbox, confidence, class_id=0, embedding=None):
        self.bbox, self.confidence, self.class_id, self.embedding = bbox, confidence, class_id, embedding


class ByteTracker:
    """bytetrack.py:35-316: high-confidence detections against all live tracks (IoU >= match_thresh), low-confidence ones
    against the tracks that left over (IoU >= 0.5), left-over high-confidence ones against LOST tracks (IoU >= 0.3, appearance
    weight 0.7), then new tracks from what is still unmatched."""

    def __init__(self, high_thresh=0.6, low_thresh=0.1, match_thresh=0.8, track_buffer=30, use_appearance=True, appearance_weight=0.5,
                 max_tracks=100):
        self.high_thresh, self.low_thresh, self.match_thresh = high_thresh, low_thresh, match_thresh
        self.track_buffer, self.use_appearance, self.appearance_weight = track_buffer, use_appearance, appearance_weight
        self.max_tracks = max_tracks
        self.reset()

    def reset(self):
        self.tracks = []
        self.next_id = 0
        self.track_count = 0
        self.kalman = {}
        self.frame_id = 0

    # -- track manager (track.py:159-241)
    def active_tracks(self):
        return [t for t in self.tracks if t.state == CONFIRMED]

    def _cleanup(self):
        self.tracks = [t for t in self.tracks if t.state != DELETED]
        if len(self.tracks) > self.max_tracks:
            self.tracks.sort(key=lambda t: t.time_since_update)
            self.tracks = self.tracks[:self.max_tracks]

    def _features(self, dets, tracks):
        if not self.use_appearance:
            return None, None
        df = [d.embedding for d in dets if d.embedding is not None]
        tf = [t.get_feature() for t in tracks if t.get_feature() is not None]
        if len(df) != len(dets) or len(tf) != len(tracks):
            return None, None  # IoU only (bytetrack.py:179-182)
        return np.array(df), np.array(tf)

    def _associate(self, dets, tracks, thr, appearance, weight):
        if len(dets) == 0 or len(tracks) == 0:
            return np.empty((0, 2), dtype=int), np.arange(len(dets)), np.arange(len(tracks))
        df, tf = self._features(dets, tracks) if appearance else (None, None)
        return associate_detections_to_tracks(np.array([d.bbox for d in dets]), np.array([t.bbox for t in tracks]), thr, df, tf, weight)

    def _predict_all(self):
        for t in self.tracks:
            kf = self.kalman.get(t.track_id)
            if kf is not None:
                t.predict(kf.predict())

    def _update_track(self, track, det, frame_idx):
        track.update(det.bbox, det.confidence, det.embedding, frame_idx)
        kf = self.kalman.get(track.track_id)
        if kf is not None:
            kf.update(det.bbox)

    def update(self, detections, frame_idx=None):
        if frame_idx is None:
            frame_idx = self.frame_id
        self.frame_id = frame_idx + 1
        if len(detections) == 0:
            self._predict_all()
            for t in self.tracks:
                t.mark_missed()
            return self.active_tracks()
        high = [d for d in detections if d.confidence >= self.high_thresh]
        low = [d for d in detections if self.low_thresh <= d.confidence < self.high_thresh]
        live = [t for t in self.tracks if t.state != DELETED]
        self._predict_all()
        m_h, un_d_h, un_t_h = self._associate(high, live, self.match_thresh, True, self.appearance_weight)
        for d, t in m_h:
            self._update_track(live[t], high[d], frame_idx)
        left = [live[i] for i in un_t_h]
        m_l, _, _ = self._associate(low, left, 0.5, False, 0.5)
        for d, t in m_l:
            self._update_track(left[t], low[d], frame_idx)
        lost = [t for t in self.tracks if t.state == LOST]
        un_high = [high[i] for i in un_d_h]
        m_r, still, _ = self._associate(un_high, lost, 0.3, True, 0.7)
        for d, t in m_r:
            self._update_track(lost[t], un_high[d], frame_idx)
        reactivated = [lost[t] for _, t in m_r]
        for t in left:
            if not any(t is r for r in reactivated):
                t.mark_missed()
        for i in still:
            det = un_high[i]
            track = Track(self.next_id, det.bbox, det.confidence, det.embedding, frame_idx)
            self.next_id += 1
            self.tracks.append(track)
            self.track_count += 1
            self.kalman[track.track_id] = KalmanBoxTracker(det.bbox)
        self._cleanup()
        return self.active_tracks()

    def get_statistics(self):
        return {"total_tracks": self.track_count, "active_tracks": len(self.active_tracks()),
                "confirmed": len([t for t in self.tracks if t.state == CONFIRMED]),
                "tentative": len([t for t in self.tracks if t.state == TENTATIVE]),
                "lost": len([t for t in self.tracks if t.state == LOST]), "frame_id": self.frame_id,
                "high_thresh": self.high_thresh, "low_thresh": self.low_thresh, "use_appearance": self.use_appearance}


def parse_yolo_detections(yolo_data):
    """main.py:236-266: the three shapes a YOLO result file may have -> {frame: [detection dicts]}."""
    by_frame = {}
    if "detections" in yolo_data:
        for item in yolo_data["detections"]:
            frame = item.get("frame", 0)
            by_frame.setdefault(frame, [])
            if "detections" in item and isinstance(item["detections"], list):
                by_frame[frame].extend(item["detections"])
            elif "bbox" in item:
                by_frame[frame].append(item)
    elif "frames" in yolo_data:
        for fd in yolo_data["frames"]:
            by_frame[fd.get("frame_number", 0)] = fd.get("detections", [])
    elif "frame_results" in yolo_data:
        for k, v in yolo_data["frame_results"].items():
            by_frame[int(k)] = v
    return by_frame


def track_video(yolo_data, tracker=None):
    """main.py:159-213 without the I/O: (frame_tracks, track_summaries, statistics) of one YOLO result."""
    tracker = tracker or ByteTracker()
    tracker.reset()
    by_frame = parse_yolo_detections(yolo_data)
    frame_tracks = []
    for frame_idx in sorted(by_frame.keys()):
        dets = [Detection(np.array(d["bbox"]), d["confidence"], d.get("class_id", 0), None) for d in by_frame[frame_idx]]
        for t in tracker.update(dets, frame_idx):
            frame_tracks.append({"frame": frame_idx, "track_id": t.track_id, "bbox": t.bbox.tolist(), "confidence": t.confidence,
                                 "state": t.state})
    summaries = []
    for t in tracker.tracks:
        if t.hits >= 3:
            summaries.append({"track_id": t.track_id, "start_frame": t.frame_history[0] if t.frame_history else 0,
                              "end_frame": t.frame_history[-1] if t.frame_history else 0, "total_frames": len(t.frame_history),
                              "avg_confidence": sum(b[4] if len(b) > 4 else t.confidence for b in t.bbox_history) / max(1, len(t.bbox_history))})
    return by_frame, frame_tracks, summaries, tracker.get_statistics()


class MemoryIdentityStore:
    """The slice of Qdrant reid/matcher.py uses (COSINE collection `cow_identities`: query top-k with payload, retrieve with
    vector, upsert, point count), in memory — for tests and single-process deployments."""

    def __init__(self):
        self.points = {}

    def ensure_collection(self, dim):
        self.dim = dim

    def count(self):
        return len(self.points)

    def query(self, vector, top_k=5):
        q = np.asarray(vector, np.float64)
        out = []
        for pid, (v, payload) in self.points.items():
            out.append((pid, float(np.dot(q, v) / (np.linalg.norm(q) * np.linalg.norm(v) + 1e-30)), payload))
        out.sort(key=lambda r: -r[1])
        return out[:top_k]

    def retrieve(self, pid):
        return self.points.get(pid)

    def upsert(self, pid, vector, payload):
        self.points[pid] = (np.asarray(vector, np.float64), dict(payload))


class QdrantIdentityStore:
    """The same four calls on qdrant_client (matcher.py:79-98,121-126,242-244,257-262) when the package and a server exist."""

    COLLECTION = "cow_identities"

    def __init__(self, url):
        from qdrant_client import QdrantClient  # ImportError offline

        self.client = QdrantClient(url=url)

    def ensure_collection(self, dim):
        from qdrant_client.http.models import Distance, VectorParams

        if self.COLLECTION not in [c.name for c in self.client.get_collections().collections]:
            self.client.create_collection(collection_name=self.COLLECTION, vectors_config=VectorParams(size=dim, distance=Distance.COSINE))

    def count(self):
        return self.client.get_collection(self.COLLECTION).points_count

    def query(self, vector, top_k=5):
        res = self.client.query_points(collection_name=self.COLLECTION, query=list(map(float, vector)), limit=top_k, with_payload=True)
        return [(str(p.id), float(p.score), p.payload) for p in res.points]

    def retrieve(self, pid):
        pts = self.client.retrieve(collection_name=self.COLLECTION, ids=[pid], with_vectors=True)
        return (np.array(pts[0].vector), pts[0].payload) if pts else None

    def upsert(self, pid, vector, payload):
        from qdrant_client.http.models import PointStruct

        self.client.upsert(collection_name=self.COLLECTION, points=[PointStruct(id=pid, vector=list(map(float, vector)), payload=payload)])


class CowReIDMatcher:
    """reid/matcher.py:40-324: best identity by cosine similarity; >= 0.75 updates that identity's vector with momentum 0.9,
    otherwise a new identity COW-%04d is created (auto_create_identities)."""

    HIGH, MEDIUM, LOW = 0.85, 0.75, 0.65

    def __init__(self, store, embedding_dim=768, auto_create_identities=True, embedding_momentum=0.9, new_uuid=uuid.uuid4):
        self.store, self.embedding_dim = store, embedding_dim
        self.auto_create_identities, self.embedding_momentum = auto_create_identities, embedding_momentum
        self.identity_counter = 0
        self.new_uuid = new_uuid

    def connect(self):
        self.store.ensure_collection(self.embedding_dim)
        self.identity_counter = self.store.count()

    def _confidence(self, score):
        return "high" if score >= self.HIGH else "medium" if score >= self.MEDIUM else "low" if score >= self.LOW else "none"

    def match_embedding(self, embedding, top_k=5):
        embedding = embedding / (np.linalg.norm(embedding) + 1e-8)
        cands = [{"identity_id": p["identity_id"], "cow_id": p["cow_id"], "similarity": s, "confidence": self._confidence(s),
                  "is_new_identity": False} for _, s, p in self.store.query(embedding, top_k)]
        best = cands[0] if cands and cands[0]["similarity"] >= self.LOW else None
        return best, cands

    def create_identity(self, embedding, tag_number=None, metadata=None):
        self.identity_counter += 1
        identity_id = str(self.new_uuid())
        cow_id = f"COW-{self.identity_counter:04d}"
        embedding = embedding / (np.linalg.norm(embedding) + 1e-8)
        self.store.upsert(identity_id, embedding, {"identity_id": identity_id, "cow_id": cow_id, "tag_number": tag_number,
                                                   "total_sightings": 1, **(metadata or {})})
        return identity_id, cow_id

    def _update_identity_embedding(self, identity_id, new_embedding):
        point = self.store.retrieve(identity_id)
        if point is None:
            return
        old, payload = point
        new_embedding = new_embedding / (np.linalg.norm(new_embedding) + 1e-8)
        upd = self.embedding_momentum * np.asarray(old) + (1 - self.embedding_momentum) * new_embedding
        upd = upd / (np.linalg.norm(upd) + 1e-8)
        self.store.upsert(identity_id, upd, {**payload, "total_sightings": payload.get("total_sightings", 0) + 1})

    def match_or_create(self, embedding, video_id, track_id, metadata=None):
        best, cands = self.match_embedding(embedding)
        if best is not None and best["similarity"] >= self.MEDIUM:
            self._update_identity_embedding(best["identity_id"], embedding)
            return best
        if self.auto_create_identities:
            identity_id, cow_id = self.create_identity(embedding, None, {"first_video": video_id, "first_track": track_id, **(metadata or {})})
            return {"identity_id": identity_id, "cow_id": cow_id, "similarity": 1.0, "confidence": "high", "is_new_identity": True}
        return {"identity_id": str(self.new_uuid()), "cow_id": "UNKNOWN", "similarity": cands[0]["similarity"] if cands else 0.0,
                "confidence": "low", "is_new_identity": True}


def video_embedding(dinov3_data):
    """main.py:291-308: `embedding`, else the mean of `canonical_frames[*].embedding`, else `video_embedding`."""
    if "embedding" in dinov3_data:
        return np.array(dinov3_data["embedding"])
    if "canonical_frames" in dinov3_data and dinov3_data["canonical_frames"]:
        embs = [np.array(f["embedding"]) for f in dinov3_data["canonical_frames"] if "embedding" in f]
        return np.mean(embs, axis=0) if embs else None
    if "video_embedding" in dinov3_data:
        return np.array(dinov3_data["video_embedding"])
    return None


class TrackingService:
    """services/tracking-service/app/main.py TrackingService, minus the relational database."""

    def __init__(self, bus, identity_store, cfg=None, results_dir="/app/data/results/tracking"):
        self.bus, self.cfg = bus, cfg or {}
        self.trackers = {}
        self.reid_matcher = CowReIDMatcher(identity_store, embedding_dim=768)
        self.results_dir = results_dir
        os.makedirs(results_dir, exist_ok=True)
        self.pending_tracks = {}
        self.video_embeddings = {}

    def _tracker(self, video_id):
        if video_id not in self.trackers:
            self.trackers[video_id] = ByteTracker(high_thresh=0.6, low_thresh=0.1, match_thresh=0.8, track_buffer=30,
                                                  use_appearance=True, appearance_weight=0.5)
        return self.trackers[video_id]

    async def process_yolo_results(self, message):
        video_id = message.get("video_id")
        if not video_id:
            return
        print(f"Tracking service processing YOLO results for {video_id}")
        try:
            results_path = message.get("results_path")
            if results_path:
                if not os.path.exists(results_path):
                    print(f"  YOLO results file not found: {results_path}")
                    return
                with open(results_path) as f:
                    yolo_data = json.load(f)
            else:
                yolo_data = message
            by_frame, frame_tracks, summaries, stats = track_video(yolo_data, self._tracker(video_id))
            if not by_frame:
                print("  No detections found in YOLO results")
                return
            self.pending_tracks[video_id] = summaries
            results = {"video_id": video_id, "pipeline": "tracking", "total_tracks": len(summaries), "track_summaries": summaries,
                       "frame_tracks": frame_tracks, "statistics": stats}
            results_file = os.path.join(self.results_dir, f"{video_id}_tracking.json")
            with open(results_file, "w") as f:
                json.dump(results, f, indent=2)
            await self.bus.publish("tracking.complete", {"video_id": video_id, "results_path": str(results_file),
                                                         "total_tracks": len(summaries), "pending_reid": True})
            print(f"  Tracking complete: {len(summaries)} tracks detected")
        except Exception as e:  # the reference logs and carries on (main.py:230-233)
            print(f"  Error in tracking: {e}")
            import traceback

            traceback.print_exc()

    async def process_dinov3_results(self, message):
        video_id = message.get("video_id")
        if not video_id:
            return
        print(f"Tracking service processing DINOv3 results for {video_id}")
        try:
            results_path = message.get("results_path")
            embedding = None
            if results_path and os.path.exists(results_path):
                with open(results_path) as f:
                    embedding = video_embedding(json.load(f))
            if embedding is None or len(embedding) == 0:
                print(f"  No embedding found for {video_id}")
                return
            self.video_embeddings[video_id] = embedding
            if video_id in self.pending_tracks:
                await self._perform_reid(video_id, embedding)
        except Exception as e:
            print(f"  Error processing DINOv3 for Re-ID: {e}")
            import traceback

            traceback.print_exc()

    async def _perform_reid(self, video_id, embedding):
        pending = self.pending_tracks.get(video_id, [])
        if not pending:
            return
        print(f"  Performing Re-ID for {len(pending)} tracks")
        reid_results = []
        for track in pending:  # the clip's embedding stands for every track of the clip (main.py:333-343)
            m = self.reid_matcher.match_or_create(embedding, video_id, track["track_id"],
                                                  {"start_frame": track["start_frame"], "end_frame": track["end_frame"]})
            reid_results.append({"track_id": track["track_id"], "cow_id": m["cow_id"], "identity_id": str(m["identity_id"]),
                                 "similarity": m["similarity"], "confidence": m["confidence"], "is_new": m["is_new_identity"]})
        results_file = os.path.join(self.results_dir, f"{video_id}_tracking.json")
        if os.path.exists(results_file):
            with open(results_file) as f:
                results = json.load(f)
            results["reid_results"] = reid_results
            results["reid_complete"] = True
            with open(results_file, "w") as f:
                json.dump(results, f, indent=2)
        await self.bus.publish("tracking.reid.match", {"video_id": video_id, "matches": reid_results,
                                                       "new_identities": sum(1 for r in reid_results if r["is_new"])})
        print(f"  Re-ID complete: {sum(1 for r in reid_results if not r['is_new'])} matched, "
              f"{sum(1 for r in reid_results if r['is_new'])} new")
        del self.pending_tracks[video_id]

    async def start(self):
        await self.bus.connect()
        self.reid_matcher.connect()
        subjects = self.cfg.get("nats", {}).get("subjects", {})
        await self.bus.subscribe(subjects.get("pipeline_yolo", "pipeline.yolo"), self.process_yolo_results)
        await self.bus.subscribe(subjects.get("pipeline_dinov3", "pipeline.dinov3"), self.process_dinov3_results)
