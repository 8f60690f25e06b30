"""DINOv3Pipeline — mirror of services/dinov3-pipeline/app/main.py: one embedding per max(1, int(fps))-th frame (mean
over ALL tokens), canonical frames [first, middle, last], float64 clip mean -> vector-store search (top 5) ->
neighbor_evidence -> upsert, `{id}_dinov3.json` + `pipeline.dinov3` (Appendix B.3).  `processor(...)` + `model(**inputs)`
(main.py:107-113) are replaced by one batched ``DinoEmbedder.embed_frames`` over the sampled frames."""
import json
import traceback
from pathlib import Path

import numpy as np
import torch

from . import runtime as R


class DINOv3Pipeline:
    def __init__(self, embedder, bus, store=None, config=None, results_dir="/app/data/results/dinov3", batch=64):
        self.config = config or R.load_config()
        self.nats_client = bus
        self.model = embedder
        self.store = store if store is not None else R.MemoryVectorStore()
        self.collection_name = self.config.get("qdrant", {}).get("collection_name", "cow_embeddings")
        self.results_dir = Path(results_dir)
        self.results_dir.mkdir(parents=True, exist_ok=True)
        self.batch = batch
        try:
            self.store.ensure_collection(embedder.cfg.hidden)
        except Exception as e:  # noqa: BLE001 — main.py:92-93 prints and carries on
            print(f"Error ensuring collection: {e}")

    @staticmethod
    def embeddings_result(embs, total, fps):
        """main.py:143-163: the embedding list plus canonical frames [first, middle, last]."""
        canonical = [embs[0], embs[len(embs) // 2], embs[-1]] if embs else []
        return {"embeddings": embs, "canonical_frames": canonical, "total_frames": total, "fps": fps}

    def extract_video_embeddings(self, video_path):
        clip = R.Clip.open(video_path)
        fps, total = clip.fps, clip.total_frames
        embs = []
        dev = self.model.device
        for chunk, host in clip.batches(max(1, fps), self.batch):
            e = self.model.embed_frames(torch.from_numpy(host).to(dev)).cpu().numpy()
            for fid, v in zip(chunk, e):
                embs.append({"frame": fid, "time": fid / fps if fps > 0 else 0, "embedding": v.tolist()})
        return self.embeddings_result(embs, total, fps)

    def search_similar(self, query, top_k=5):
        try:
            return self.store.search(query, top_k)
        except Exception as e:  # noqa: BLE001
            print(f"Error searching similar: {e}")
            return []

    async def process_video(self, video_data):
        video_id = video_data["video_id"]
        processed_path = Path(video_data["processed_path"])
        if not processed_path.exists():
            print(f"Processed video not found: {processed_path}")
            return
        try:
            await self.finish(video_data, self.extract_video_embeddings(processed_path))
        except Exception as e:  # noqa: BLE001
            print(f"Error in DINOv3 pipeline for {video_id}: {e}")
            traceback.print_exc()

    async def finish(self, video_data, data):
        """main.py:197-275 after the embeddings exist: float64 clip mean, top-5 search, neighbor_evidence, upsert, file, publish."""
        video_id = video_data["video_id"]
        if not data["embeddings"]:
            print(f"No embeddings extracted for {video_id}")
            return
        avg = np.mean([np.array(e["embedding"]) for e in data["embeddings"]], axis=0)  # float64 (Appendix C-9)
        similar = self.search_similar(avg, top_k=5)
        evidence = 0.5
        if similar:
            labels = [c["label"] for c in similar if c["label"] is not None]
            if labels:
                evidence = sum(1 for lab in labels if lab == 1) / len(labels)
        try:
            self.store.upsert(video_id, avg.tolist(), {"video_id": video_id, "filename": video_data.get("filename", ""),
                                                       "uploaded_at": video_data.get("uploaded_at", ""), "label": None,
                                                       "metadata": video_data.get("metadata", {})})
        except Exception as e:  # noqa: BLE001
            print(f"Error storing in VectorDB: {e}")
        results = {"video_id": video_id, "embedding_dim": len(avg), "num_embeddings": len(data["embeddings"]),
                   "similar_cases": similar, "neighbor_evidence": evidence, "canonical_frames": data["canonical_frames"]}
        results_file = self.results_dir / f"{video_id}_dinov3.json"
        with open(results_file, "w") as f:
            json.dump(results, f, indent=2)
        await self.nats_client.publish(self.config["nats"]["subjects"]["pipeline_dinov3"], {
            "video_id": video_id, "pipeline": "dinov3", "results_path": str(results_file), "neighbor_evidence": evidence,
            "similar_cases": similar, "embedding_dim": len(avg)})

    async def start(self):
        await self.nats_client.connect()
        await self.nats_client.subscribe(self.config["nats"]["subjects"]["video_preprocessed"], self.process_video)
