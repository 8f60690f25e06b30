"""Multi-GPU: one process per GPU, the frames of a clip sharded as contiguous blocks, and ONE collective per clip (or bench
step) to reassemble the per-frame records (RCCL all-gather or gather-to-rank-0 over xGMI; the `nccl` backend
IS RCCL on ROCm, `gloo` in the CPU tests).  There is no data-path collective inside the networks.

A record is a fixed-stride byte row per frame: {valid, count, boxes, scores, classes, mask statistics, mask bits, embedding, ...}
— every field of the step's output dict viewed as bytes, 8-byte aligned, concatenated.  Shards are padded to ceil(n / world)
rows (padding rows carry valid = 0), so every rank contributes the same number of bytes whatever n % world is."""
import math

import torch
import torch.distributed as dist


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of rank: ceil(n/world) frames per rank, so per-clip order is a plain concat."""
    per = -(-n_frames // world)
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def shard_rows(n_frames, world):
    """Rows every rank contributes to the gather (the last shards may be short or empty and are padded up to this)."""
    return -(-n_frames // world) if n_frames > 0 else 0


def record_layout(rec):
    """[(name, dtype, tail shape, byte offset, bytes)] + row stride for a dict of per-frame tensors [n, ...].
    Row byte 0..7 is the int64 `valid` flag."""
    off, layout = 8, []
    for name in sorted(rec):
        v = rec[name]
        nb = v.element_size() * math.prod(int(d) for d in v.shape[1:])
        layout.append((name, v.dtype, tuple(v.shape[1:]), off, nb))
        off += -(-nb // 8) * 8
    return layout, off


def pack_records(rec, n_rows=None):
    """dict of per-frame tensors [n, ...] (same n, same device) -> (uint8 [n_rows, stride], layout); rows >= n are padding."""
    names = sorted(rec)
    n = int(rec[names[0]].shape[0])
    for k in names:
        if int(rec[k].shape[0]) != n:
            raise ValueError(f"pack_records: field {k} has {rec[k].shape[0]} rows, expected {n}")
    n_rows = n if n_rows is None else int(n_rows)
    if n_rows < n:
        raise ValueError(f"pack_records: {n} records do not fit {n_rows} rows")
    layout, stride = record_layout(rec)
    dev = rec[names[0]].device
    buf = torch.zeros((n_rows, stride), dtype=torch.uint8, device=dev)
    if n:
        buf[:n, :8] = torch.ones((n, 1), dtype=torch.int64, device=dev).view(torch.uint8)
        for name, dtype, tail, off, nb in layout:
            buf[:n, off:off + nb] = rec[name].contiguous().view(n, -1).view(torch.uint8)
    return buf, layout


def unpack_records(buf, layout, keep_valid_only=True):
    """uint8 [rows, stride] -> dict of tensors; padding rows (valid = 0) are dropped."""
    valid = buf[:, :8].contiguous().view(torch.int64)[:, 0] != 0
    rows = buf[valid] if keep_valid_only else buf
    out = {}
    for name, dtype, tail, off, nb in layout:
        out[name] = rows[:, off:off + nb].contiguous().view(dtype).view((rows.shape[0],) + tuple(tail))
    return out


def check_equal_rows(n_rows, device=None):
    """Every rank must contribute the same number of rows, or the collective hangs (RCCL) / errors (gloo): one cheap
    all-reduce of (min, -max) catches a caller that forgot to pad.  `device`: where the records live (RCCL reduces in HBM of
    THAT device, not of the process's current one)."""
    t = torch.tensor([n_rows, -n_rows], dtype=torch.int64)
    if dist.get_backend() != "gloo":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    lo, hi = int(t[0]), -int(t[1])
    if lo != hi:
        raise RuntimeError(f"gather: ranks contribute between {lo} and {hi} rows; pad shards with shard_rows()")


def gather_packed(buf, root=None, check=False):
    """ONE collective: uint8 [rows, stride] per rank -> [world * rows, stride] on `root` (None elsewhere), or on every rank
    when root is None (all-gather).  gloo moves host tensors (CPU tests, single-GPU rehearsals); nccl = RCCL moves HBM."""
    world = dist.get_world_size()
    if check:
        check_equal_rows(int(buf.shape[0]), buf.device)
    dev = buf.device
    via_host = dist.get_backend() == "gloo" and dev.type != "cpu"
    src = buf.contiguous().cpu() if via_host else buf.contiguous()
    if root is None:
        g = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(g, src)
        return g.to(dev) if via_host else g
    if dist.get_rank() == root:
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.gather(src, parts, dst=root)
        g = torch.cat(parts, 0)
        return g.to(dev) if via_host else g
    dist.gather(src, None, dst=root)
    return None


def gather_clip_records(rec, n_total, root=0):
    """The per-clip exchange of §8e: this rank's records (its shard_range block of a clip of `n_total` sampled frames, possibly
    empty) -> on `root` the dict for all n_total frames in clip order; None on the other ranks.  Single-process: returns rec."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    buf, layout = pack_records(rec, shard_rows(n_total, world))
    g = gather_packed(buf, root=root)
    if g is None:
        return None
    out = unpack_records(g, layout)
    n = int(next(iter(out.values())).shape[0])
    if n != n_total:
        raise RuntimeError(f"gather_clip_records: {n} valid records arrived, the clip has {n_total}")
    return out


def gather_frame_records(rec):
    """All-gather form (every rank ends up with every record; bench.py's multi-rank step): ONE all_gather_into_tensor of the
    packed rows.  Ranks must hold equally many records (bench shards are equal by construction; otherwise use
    gather_clip_records, which pads)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    buf, layout = pack_records(rec)
    return unpack_records(gather_packed(buf, root=None), layout)
