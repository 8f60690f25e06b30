"""Multi-GPU plumbing on the CPU: world_size-2 gloo runs of lmx.dist (contiguous frame shards, fixed-stride packed records,
ONE collective per clip) and of the FusedFeatureService with stand-in backends — the same code paths bench.py and the
service use with the nccl (= RCCL) backend on a node."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lmx import dist as ldist

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(idx):
    idx = torch.as_tensor(idx, dtype=torch.int64)
    n = len(idx)
    return {"boxes": idx.float().view(-1, 1, 1).expand(n, 3, 4).contiguous(), "counts": idx.int(),
            "embedding": idx.float().view(-1, 1).expand(n, 8).contiguous(), "mask_bits": (idx % 251).to(torch.uint8).view(-1, 1, 1).expand(n, 5, 3).contiguous(),
            "mask_stats": idx.view(-1, 1).expand(n, 8).contiguous() * 1000003}


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _count_collectives():
    """Wrap the data collectives lmx.dist may use and count the calls."""
    calls = []
    for name in ("all_gather_into_tensor", "gather", "all_gather", "broadcast", "all_to_all"):
        real = getattr(dist, name)
        setattr(dist, name, (lambda real, name: lambda *a, **k: (calls.append(name), real(*a, **k))[1])(real, name))
    return calls


def _worker_gather(rank, world, port, q, n_frames):
    _init(rank, world, port)
    try:
        calls = _count_collectives()
        lo, hi = ldist.shard_range(n_frames, rank, world)
        rec = _records(range(lo, hi))            # the last shard is SHORT (or empty): the library pads, not the caller
        out = ldist.gather_clip_records(rec, n_frames, root=0)
        ok = True
        if rank == 0:
            want = _records(range(n_frames))
            ok = set(out) == set(want) and all(torch.equal(out[k], want[k]) and out[k].dtype == want[k].dtype for k in want)
        else:
            ok = out is None
        q.put((rank, bool(ok), list(calls), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_contiguous_partition():
    for n, w in [(150, 8), (150, 1), (7, 2), (3, 8), (32, 4), (0, 4)]:
        parts = [ldist.shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert max(hi - lo for lo, hi in parts) == ldist.shard_rows(n, w)


def test_pack_unpack_round_trip_and_padding():
    rec = _records(range(5))
    buf, layout = ldist.pack_records(rec, 8)
    assert buf.shape[0] == 8 and buf.dtype == torch.uint8 and buf.shape[1] % 8 == 0
    back = ldist.unpack_records(buf, layout)
    assert all(torch.equal(back[k], rec[k]) for k in rec)
    empty, layout0 = ldist.pack_records({k: v[:0] for k, v in rec.items()}, 3)
    assert layout0 == layout and int(empty.sum()) == 0 and ldist.unpack_records(empty, layout)["counts"].shape == (0,)
    with pytest.raises(ValueError):
        ldist.pack_records(rec, 4)


@pytest.mark.parametrize("n_frames", [7, 1, 10])
def test_gather_clip_records_world2_one_collective(n_frames):
    res = _spawn(_worker_gather, 2, n_frames)
    assert [r[1] for r in res] == [True, True]
    for _, _, calls, _ in res:
        assert calls == ["gather"], f"expected ONE collective per clip, saw {calls}"
    per = -(-n_frames // 2)
    assert [r[3] for r in res] == [(0, min(per, n_frames)), (min(per, n_frames), n_frames)]


def _worker_allgather(rank, world, port, q, per):
    _init(rank, world, port)
    try:
        calls = _count_collectives()
        out = ldist.gather_frame_records(_records(range(rank * per, (rank + 1) * per)))
        want = _records(range(world * per))
        q.put((rank, all(torch.equal(out[k], want[k]) for k in want), list(calls), None))
    finally:
        dist.destroy_process_group()


def test_gather_frame_records_world2_one_collective():
    res = _spawn(_worker_allgather, 2, 4)
    assert [r[1] for r in res] == [True, True]
    assert all(r[2] == ["all_gather_into_tensor"] for r in res)


def _worker_unequal(rank, world, port, q):
    _init(rank, world, port)
    try:
        buf, _ = ldist.pack_records(_records(range(3 + rank)))
        try:
            ldist.gather_packed(buf, root=None, check=True)
            q.put((rank, False, [], None))
        except RuntimeError as e:
            q.put((rank, "pad shards" in str(e), [], None))
    finally:
        dist.destroy_process_group()


def test_unequal_rows_are_refused_before_the_collective():
    assert [r[1] for r in _spawn(_worker_unequal, 2)] == [True, True]


def _worker_service(rank, world, port, q, root_dir, clip_path, schedule):
    sys.path.insert(0, HERE)
    import asyncio

    import test_services_host as H
    from lmx import services
    from lmx.services import runtime as R

    if world > 1:
        _init(rank, world, port)
    try:
        calls = _count_collectives() if world > 1 else []
        from pathlib import Path

        bus = R.InProcessBus()
        per_frame = (lambda fid: [([2 + fid % 5, 2, 20 + fid % 7, 22], 0.9, 19)] if fid not in (30, 60) else [])
        _, _, _, fused, fx = H._three_and_fused(Path(root_dir), f"w{world}r{rank}", bus, per_frame, schedule, chunk=3)
        asyncio.run(fused.process_video({"video_id": "z", "processed_path": clip_path, "filename": "z.mp4"}))
        q.put((rank, [p[0] for p in bus.published], list(calls), sum(n for n, _, _ in fx.steps)))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("schedule,n_frames", [("reference", 100), ("dense", 41)])
def test_fused_service_world2_equals_world1(tmp_path, schedule, n_frames):
    """The service at world size 2 (frames split in two contiguous blocks, ONE gather of packed records, rank 0 writes and
    publishes) produces byte-identical JSONs to the single-process run; n_frames and the sampled counts are odd on purpose."""
    sys.path.insert(0, HERE)
    import test_services_host as H

    clip = H._clip(tmp_path, n_frames, 30.0)
    one = _spawn(_worker_service, 1, str(tmp_path), str(clip), schedule)
    two = _spawn(_worker_service, 2, str(tmp_path), str(clip), schedule)
    assert one[0][1] == ["pipeline.yolo", "pipeline.sam3", "pipeline.dinov3"]
    assert two[0][1] == ["pipeline.yolo", "pipeline.sam3", "pipeline.dinov3"] and two[1][1] == [], "only rank 0 publishes"
    for _, _, calls, _ in two:
        assert [c for c in calls if c != "all_reduce"] == ["gather"], calls
    assert two[0][3] + two[1][3] == one[0][3] and two[0][3] > 0 and two[1][3] > 0, "frames were not split over the ranks"
    for sub, key in (("yolo", "yolo"), ("sam3", "sam3"), ("dino", "dinov3")):
        a = open(tmp_path / "w1r0" / sub / f"z_{key}.json").read()
        b = open(tmp_path / "w2r0" / sub / f"z_{key}.json").read()
        assert a == b, f"{key} JSON differs between world 1 and world 2"
        assert not (tmp_path / "w2r1" / sub / f"z_{key}.json").exists()
    assert json.loads(a)["num_embeddings"] == len(range(0, n_frames, 30))


def _worker_failing_rank(rank, world, port, q, root_dir, clip_path, how):
    """One rank's local pass fails (extractor raises / its clip is missing); both ranks must come back."""
    sys.path.insert(0, HERE)
    import asyncio
    import time
    from pathlib import Path

    import test_services_host as H
    from lmx.services import runtime as R

    _init(rank, world, port)
    try:
        calls = _count_collectives()
        bus = R.InProcessBus()
        per_frame = (lambda fid: [([2 + fid % 5, 2, 20 + fid % 7, 22], 0.9, 19)])
        _, _, _, fused, fx = H._three_and_fused(Path(root_dir), f"fail_{how}_r{rank}", bus, per_frame, "reference", chunk=3)
        path = clip_path
        if rank == 1 and how == "raise":
            def boom(*a, **k):
                raise RuntimeError("injected failure on rank 1")
            fx.step = boom
        if rank == 1 and how == "missing":
            path = clip_path + ".not_on_this_rank"
        t0 = time.time()
        asyncio.run(fused.process_video({"video_id": "z", "processed_path": path, "filename": "z.mp4"}))
        q.put((rank, [p[0] for p in bus.published], list(calls), time.time() - t0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("how", ["raise", "missing"])
def test_fused_service_world2_failing_rank_does_not_hang(tmp_path, how):
    """A rank whose local pass fails (exception in the extractor, or the clip missing on its filesystem) still joins the
    control all-reduce, which carries an error flag: BOTH ranks skip the gather together, return within the timeout, and
    nothing is written or published — the single-process error convention (yolo main.py:203-206)."""
    sys.path.insert(0, HERE)
    import test_services_host as H

    clip = H._clip(tmp_path, 64, 30.0)
    res = _spawn(_worker_failing_rank, 2, str(tmp_path), str(clip), how)  # _spawn's q.get(timeout) is the hang detector
    for rank, published, calls, secs in res:
        assert published == [], f"rank {rank} published {published} although a rank failed"
        assert "gather" not in calls and calls.count("all_reduce") <= 1, calls
        assert secs < 60
    assert not list(tmp_path.glob("fail_*/*/z_*.json"))
