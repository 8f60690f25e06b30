"""Multi-GPU plumbing on the CPU: world_size-2 gloo run of lmx.dist (contiguous frame shards, one all_gather per field),
the same code path bench.py / the fused service use with the nccl (= RCCL) backend on a node."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lmx import dist as ldist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = ldist.shard_range(n_frames, rank, world)
        per = -(-n_frames // world)
        # every rank holds `per` records (the last shard is padded), tagged with the global frame index
        idx = torch.arange(lo, lo + per)
        rec = {"boxes": idx.float().view(-1, 1, 1).expand(-1, 3, 4).contiguous(), "counts": idx.int(),
               "embedding": idx.float().view(-1, 1).expand(-1, 8).contiguous()}
        out = ldist.gather_frame_records(rec)
        ok = all(torch.equal(out["counts"][:n_frames], torch.arange(n_frames).int()) for _ in (0,))
        ok = ok and out["boxes"].shape == (world * per, 3, 4) and torch.equal(out["embedding"][:, 0], out["counts"].float())
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_contiguous_partition():
    for n, w in [(150, 8), (150, 1), (7, 2), (3, 8), (32, 4)]:
        parts = [ldist.shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert max(hi - lo for lo, hi in parts) == -(-n // w)


def test_gather_frame_records_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [(0, 4), (4, 7)]
