"""Multi-GPU: one process per GPU, frames sharded as contiguous blocks, ONE collective per clip-shard to reassemble the
per-frame records on every rank (RCCL all-gather over xGMI; `nccl` backend = RCCL on ROCm, `gloo` in the
CPU tests).  Records are fixed-stride so the gather is a single all_gather_into_tensor per field."""
import torch
import torch.distributed as dist


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of rank: ceil(n/world) frames per rank, so per-clip order is a plain concat."""
    per = -(-n_frames // world)
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def gather_frame_records(rec):
    """rec: dict of per-frame tensors [n_local, ...] with identical n_local on every rank -> dict of [world*n_local, ...].
    Boxes / scores / classes / counts, masks (+ their statistics) and DINO embeddings are what the services persist."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    # gloo (CPU tests, single-GPU rehearsals of the multi-rank bench) gathers host tensors; nccl = RCCL gathers in place
    via_host = dist.get_backend() == "gloo"
    out = {}
    for k, v in rec.items():
        dev = v.device
        v = v.contiguous()
        if via_host and dev.type != "cpu":
            v = v.cpu()
        g = torch.empty((world * v.shape[0],) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        dist.all_gather_into_tensor(g, v)
        out[k] = g.to(dev) if g.device != dev else g
    return out
