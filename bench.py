#!/usr/bin/env python3
"""bench.py — frames/sec of the fused YOLO + SAM(Hiera-B+) + DINOv3 feature-extraction path on synthetic 1080p clips
(BASELINE.json metric, cfg#5).  One process per GPU; a step = one pass of the hot path over ONE synthetic 5 s @ 30 fps
1080p clip (150 unique frames) already resident in HBM, including the pack of the per-frame records
(boxes, scores, classes, bit-packed mask + statistics, embedding), the one gather to rank 0 when there are several ranks,
and their copy to pinned host memory — what the services persist.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` is the DENSE schedule (every frame through all
three networks: the throughput mode, YOLO on its f16 plan); `reference_schedule` reports the services' own schedule (YOLO +
SAM on frames 0, 15, ..., 135, DINO on 0, 30, ..., 120: yolo main.py:67, dinov3 main.py:127) on the same clip with YOLO on its
EXACT plan (keep-sets of the fp32 path: the mode whose JSON must match).
With N > 1 ranks the headline is the path north_star describes — ONE 150-frame clip per step, rank r runs its contiguous block
of ceil(150 / N) frames, one gather of the packed records to rank 0, D2H there (`"scaling": "strong"`) — and the clip-per-GPU
rate (`"weak_scaling"`: every rank its own clip, the round-1/2 definition) is reported beside it; `--clip-per-gpu` makes the
weak form the headline.  `per_config` holds standalone rates of BASELINE cfg#2 / #3 / #4 at their stated shapes.  `roofline` is the kernel class
with the largest share of GPU time, `roofline_classes` lists every class — each timed live with HIP events on the launch
stream during K more serialized steps, bound chosen by arithmetic intensity; `cpu_baseline` is the fp32 oracle on the host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F16_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0     # HBM3E peak, same table (6.3 TB/s is what a streaming copy achieves)
# algorithmic FLOPs per 1080p frame (SURVEY.md §8d cfg#5): YOLOv8-l @384x640 + Hiera-B+ trunk+FPN @1024^2 + DINOv3 ViT-L/16 @224^2
GFLOP_PER_FRAME = {"yolo": 99.1, "sam": 645.0, "dino": 125.7}
# rocprofv3 --pmc passes of THIS command (tools/profile_round.sh -> tools/pmc.sh), the newest committed round
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_summary.json") for r in (3, 2)) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r03_pmc_summary.json"))


def cpu_baseline(n_frames, clip_seed):
    """The reference services' CPU path = the fp32 oracle (plain PyTorch CPU, batch 1, frame after frame, like the
    loops at yolo main.py:69-105, sam3 main.py:192-232, dinov3 main.py:133-146) on the same synthetic frames."""
    from lmx import dino, sam, synth, weights, yolo
    from oracle import hiera as OH
    from oracle import preprocess as OP
    from oracle import sam_decoder as OD
    from oracle import vit as OV
    from lmx import sam_decoder
    from oracle import yolo as OY

    # the box's CPU share, not the host's core count (a cgroup-limited box oversubscribed 8x runs 100x slower)
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(share, torch.get_num_threads(), 16)))
    ycfg, scfg, dcfg = yolo.YoloConfig("l"), sam.hiera_b_plus(), dino.dinov3_vitl16()
    bn = yolo.bn_stats_path("l")
    ysd = yolo.synthetic_state_dict(ycfg, 7, bn)
    ssd = weights.synth_state_dict(sam.param_spec(scfg), 5)
    dsd = weights.synth_state_dict(dino.param_spec(dcfg), 3)
    msd = sam_decoder.synthetic_state_dict(105)
    frames = [synth.synth_frame(clip_seed, i) for i in range(n_frames)]
    t0 = time.perf_counter()
    with torch.no_grad():
        for j, f in enumerate(frames):
            print(f"[bench] cpu_baseline frame {j + 1}/{n_frames} ({time.perf_counter() - t0:.1f}s)", file=sys.stderr, flush=True)
            OY.predict("l", 80, ysd, f, conf=0.5)
            fpn, _ = OH.encoder_forward(scfg, ssd, torch.from_numpy(OP.sam_pixel_values(f, 1024))[None])
            box = np.array([[300.0, 150.0, 1200.0, 900.0]], np.float32)  # a fixed prompt: the decoder's cost does not depend on it
            sp = OD.prompt_encode_box(msd, torch.from_numpy(OD.scale_box(box, f.shape[:2], (576, 1024))))
            low, _ = OD.mask_decode(msd, fpn[2], sp)
            OD.postprocess(low, (576, 1024), f.shape[:2])
            OV.embed(dcfg, dsd, torch.from_numpy(OP.dino_pixel_values(f))[None])
    dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"the first {n_frames} frames of the benched clip, dense schedule: YOLOv8-l + Hiera-B+ encoder + SAM mask decoder + DINOv3 ViT-L/16, fp32 PyTorch CPU, batch 1"}


def pmc_traffic_by_class():
    """HBM bytes per launch and kernel class from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc.sh -> tools/pmc_summary.py -> profiles/rNN_pmc_summary.json; FETCH_SIZE doubled per the gfx950 correction,
    + WRITE_SIZE).  Counters cannot be read inside the timed run, so this is the profiled figure; None without the file."""
    if not os.path.exists(PMC_SUMMARY):
        return {}
    with open(PMC_SUMMARY) as f:
        rows = json.load(f)
    acc = {}
    for name, r in rows.items():
        # the counters are per KERNEL NAME, the classes per launch (arithmetic intensity): the f32-output (residual-stream)
        # GEMM instantiations stand for the HBM-bound GEMM class, the f16-output ones for the MFMA-bound class — close, not
        # identical sets of launches (profiles/rNN_pmc_summary.txt has the per-kernel rows)
        if "hiera_attn" in name:
            cls = "fused attention half"
        elif "gemm2_kernel<1" in name or "gemm_kernel" in name and ", 1>" in name:
            cls = "gemm/hbm-bound"      # f32-output (residual-stream) instantiations
        elif "gemm" in name:
            cls = "gemm/mfma-bound"
        elif "attn_gp_kernel" in name or "attn_kernel<2, true, false, true" in name or "attn_kernel<2, false, false, true" in name:
            cls = "attention/mfma-bound"  # the LDS-DMA kernels: long flat sequences (Hiera's global blocks)
        elif "attn" in name:
            cls = "attention/hbm-bound"
        elif "mlp_kernel" in name:  # (the summary truncates long mangled names from the left)
            cls = "fused ln+mlp"
        elif "layernorm" in name:
            cls = "layernorm"
        else:
            continue
        a = acc.setdefault(cls, [0.0, 0.0])
        a[0] += r["launches"]
        a[1] += r["launches"] * (r["read_MB"] + r["write_MB"]) * 1e6
    return {k: (b / n if n else None) for k, (n, b) in acc.items()}


def roofline_classes(trace, wall_s):
    """Per kernel class: governing bound by arithmetic intensity (flop/byte against the MFMA/HBM ridge), achieved rate =
    algorithmic work / event-timed duration, fraction of the peak, share of the serialized step."""
    from lmx import kernels as K

    traffic = pmc_traffic_by_class()
    total = sum(r["seconds"] for r in trace.values()) or 1.0
    out = []
    for cls, r in sorted(trace.items(), key=lambda kv: -kv[1]["seconds"]):
        row = {"kernel": cls, "launches": r["launches"], "time_share": r["seconds"] / total, "avg_us": r["seconds"] / r["launches"] * 1e6}
        if r["modelled"] and r["bytes"] > 0:
            ai = r["flops"] / r["bytes"]
            mfma = ai >= K.RIDGE_FLOP_PER_BYTE
            if mfma:
                ach = r["flops"] / r["seconds"] / 1e12
                row.update(bound="mfma", achieved=ach, peak=PEAK_F16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_F16_TFLOPS)
            else:
                ach = r["bytes"] / r["seconds"] / 1e9
                row.update(bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS)
            row["flop_per_launch"] = r["flops"] / r["launches"]
            row["bytes_per_launch"] = r["bytes"] / r["launches"]
            row["traffic"] = traffic.get(cls)
        else:
            row.update(bound=None, achieved=None, peak=None, unit=None, frac=None, traffic=None)
        out.append(row)
    return out, total / wall_s if wall_s > 0 else None


def per_config_rates(fx, dev, rank, world, log, dist):
    """Standalone throughput of BASELINE cfg#2 / cfg#3 / cfg#4 at their stated shapes, a few passes each,
    inputs resident in HBM: frames/s per GPU, algorithmic TFLOP/s and the fraction of the dense f16 MFMA peak (all three are
    MFMA-bound by arithmetic intensity).  cfg#4's 256 frames are sharded over the ranks and the [256, 1024] embeddings
    all-gathered, as the config words it; cfg#2 / cfg#3 run the same batch on every rank (rank 0's rate is reported)."""
    from lmx import dist as ldist
    from lmx import synth

    def rate(fn, n_img, gflop_per_img, iters=4, sync_all=False):
        fn()
        torch.cuda.synchronize()
        if world > 1 and sync_all:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        if world > 1 and sync_all:
            dist.barrier()
        dt = (time.perf_counter() - t0) / iters
        tf = n_img * gflop_per_img / dt / 1e3
        return {"frames_per_s": n_img / dt, "ms_per_batch": dt * 1e3, "tflops": tf, "frac_of_mfma_peak": tf / PEAK_F16_TFLOPS}

    out = {}
    f2 = torch.from_numpy(synth.cfg2_frames()).to(dev)
    out["cfg2_yolov8l_640x640_b32"] = dict(rate(lambda: fx.yolo.detect(f2, conf=0.25, precision="f16"), 32, 165.1),
                                           plan="f16 (throughput)", gflop_per_img=165.1, per="GPU")
    out["cfg2_yolov8l_640x640_b32_exact"] = dict(rate(lambda: fx.yolo.detect(f2, conf=0.25, precision="exact"), 32, 165.1, iters=2),
                                                 plan="exact (fp32 keep-sets; 3x the MFMA work, tflops counts the fp32-equivalent work)",
                                                 gflop_per_img=165.1, per="GPU")
    del f2
    log("per_config: cfg#2 done")
    f3 = torch.from_numpy(np.stack([synth.synth_frame(2, i, 1024, 1024) for i in range(16)], 0)).to(dev)
    out["cfg3_hiera_bplus_1024_b16"] = dict(rate(lambda: fx.sam.encode(f3), 16, 645.0), gflop_per_img=645.0, per="GPU")
    del f3
    log("per_config: cfg#3 done")
    lo, hi = ldist.shard_range(256, rank, world)
    f4 = torch.from_numpy(np.stack([synth.synth_frame(21, i % 16) for i in range(lo, hi)], 0)).to(dev) if hi > lo else None
    D = fx.dino.cfg.hidden
    rows = ldist.shard_rows(256, world)

    def cfg4():
        e = torch.zeros((rows, D), dtype=torch.float32, device=dev)
        if f4 is not None:
            e[:hi - lo] = fx.dino.embed_frames(f4)
        if world > 1:
            g = torch.empty((world * rows, D), dtype=torch.float32, device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_gather_into_tensor(g, e.cpu() if dist.get_backend() == "gloo" else e)
        return e

    out["cfg4_dinov3_vitl16_224_b256"] = dict(rate(cfg4, 256, 125.7, sync_all=True), gflop_per_img=125.7, per="whole job",
                                              parallelism=f"256 frames sharded over {world} rank(s), all-gather of [256, {D}] f32",
                                              input="raw 1080p frames (Pillow-exact bicubic resize + crop + patchify on the device included)")
    if world > 1:
        out["cfg4_dinov3_vitl16_224_b256"]["frac_of_mfma_peak"] /= world
    log("per_config: cfg#4 done")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=150, help="frames of the synthetic clip per step (5 s @ 30 fps = 150)")
    ap.add_argument("--fps", type=int, default=30)
    ap.add_argument("--sam-chunk", type=int, default=30, help="frames per SAM encoder pass (each pass runs on its own HIP stream; 150 = 5 x 30)")
    ap.add_argument("--shard-sam-chunk", type=int, default=0, help="frames per SAM pass of a rank's block in the sharded-clip step (0: one pass up to 24 frames, else equal passes of <= 30)")
    ap.add_argument("--clip-per-gpu", action="store_true", help="N > 1: make the clip-per-GPU (weak) rate the headline instead of the sharded clip")
    ap.add_argument("--shard-clip", action="store_true", help="N > 1: sharded clip as the headline (the default; kept for explicitness)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-schedule", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-per-config", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=10)
    ap.add_argument("--shapes-out", default=None, help="write the per-shape table of the roofline leg (text) to this file")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # LMX_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend — a single-GPU rehearsal of the multi-rank control
    # flow (rendezvous, gather, barriers, max-over-ranks timing); never a measurement
    rehearse = bool(os.environ.get("LMX_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from lmx import dist as ldist
    from lmx import kernels as K
    from lmx import pipeline, synth

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log("building synthetic weights")
    fx = pipeline.FusedExtractor(dev)
    log(f"generating the synthetic clip ({args.frames} unique 1080p frames per rank)")
    # clip-per-GPU (weak) form: every rank owns one clip (seed = 100 + rank); frames are unique within the clip
    host = synth.synth_clip(100 + rank, args.frames)
    frames = torch.from_numpy(host).to(dev)
    del host
    i_det, i_emb = max(1, args.fps // 2), max(1, args.fps)
    sched = sorted(set(range(0, args.frames, i_det)) | set(range(0, args.frames, i_emb)))
    sched_frames = frames[sched].contiguous()
    det_idx = [j for j, i in enumerate(sched) if i % i_det == 0]
    emb_idx = [j for j, i in enumerate(sched) if i % i_emb == 0]
    # sharded-clip (strong) form: ONE clip (seed 100) per step, rank r keeps the frames of its contiguous block
    lo, hi = ldist.shard_range(args.frames, rank, world)
    shard = None
    if world > 1:
        shard = frames[lo:hi] if rank == 0 else torch.from_numpy(synth.synth_clip(100, hi - lo, start=lo)).to(dev) if hi > lo else frames[:0]
    # SAM pass size of a rank's block: one pass up to 24 frames (measured on one MI355X, 19-frame blocks = 8 ranks: one pass
    # 536 frames/s, two passes of 10 501, passes of 7 528, of 5 534), else equal passes of at most 30 frames
    blk = max(1, hi - lo)
    shard_chunk = args.shard_sam_chunk or (blk if blk <= 24 else -(-blk // -(-blk // 30)))
    pinned = {}

    def to_host(buf):
        key = tuple(buf.shape)
        if key not in pinned:
            pinned[key] = torch.empty(buf.shape, dtype=torch.uint8).pin_memory()
        pinned[key].copy_(buf, non_blocking=True)

    def persist(out):
        """What leaves the GPU per clip: ONE packed record buffer -> (one gather to rank 0) -> pinned host memory."""
        out = {k: v for k, v in out.items() if k != "mask"}
        buf, _ = ldist.pack_records(out)
        if world > 1:
            # ONE collective per step.  all_gather_into_tensor by default (RCCL's most travelled path; every rank then holds the
            # records, rank 0 persists them); LMX_BENCH_GATHER=root uses the gather-to-rank-0 form the service uses
            if os.environ.get("LMX_BENCH_GATHER") == "root":
                buf = ldist.gather_packed(buf, root=0)
                if buf is None:
                    return
            else:
                buf = ldist.gather_packed(buf, root=None)
                if rank != 0:
                    return
        to_host(buf)

    # the dense schedule is the THROUGHPUT mode: YOLO on its f16 plan; the reference schedule is the mode whose JSON must match
    # the reference: YOLO on its exact plan (lmx.yolo.YoloDetector)
    def step_dense():
        persist(fx.step(frames, sam_chunk=args.sam_chunk, precision="f16"))

    def step_reference():
        persist(fx.step(sched_frames, sam_chunk=args.sam_chunk, det_idx=det_idx, emb_idx=emb_idx, precision="exact"))

    def step_sharded():
        """north_star's multi-GPU path: this rank's block of THE clip, then ONE gather of the packed records to rank 0 (the
        service's collective: lmx.dist.gather_clip_records pads the short last blocks), D2H on rank 0."""
        out = {k: v for k, v in fx.step(shard, sam_chunk=shard_chunk, precision="f16").items() if k != "mask"}
        buf, _ = ldist.pack_records(out, ldist.shard_rows(args.frames, world))
        g = ldist.gather_packed(buf, root=0)
        if g is not None:
            to_host(g)

    def timed(step, label):
        for i in range(args.warmup):
            step()
            torch.cuda.synchronize()
            log(f"{label}: warmup step {i + 1}/{args.warmup} done")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        log(f"{label}: {args.steps} steps in {dt:.3f}s")
        return dt

    strong = world > 1 and not args.clip_per_gpu
    dt_shard = timed(step_sharded, "sharded clip (one clip per step over all ranks)") if world > 1 else None
    dt_dense = timed(step_dense, "dense schedule, one clip per GPU")
    dt = dt_shard if strong else dt_dense
    dt_ref = None if args.no_reference_schedule else timed(step_reference, "reference schedule")

    # Roofline leg: the timed steps keep up to seven HIP streams in flight, so an event pair around one launch would time its
    # neighbours too.  Every launch is therefore bracketed on K more steps of the SAME workload run on one stream (not part
    # of `value`); rocprofv3's per-kernel averages in profiles/ are taken the same way (LMX_SERIAL=1).
    # (every rank replays, without collectives; rank 0 reports)
    classes, traced_share, dt_serial = None, None, None
    if not args.no_roofline:
        fx.serial = True
        fx.step(frames, sam_chunk=args.sam_chunk, precision="f16")
        torch.cuda.synchronize()
        K.start_launch_trace()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            fx.step(frames, sam_chunk=args.sam_chunk, precision="f16")
        torch.cuda.synchronize()
        dt_serial = time.perf_counter() - t1
        trace, shapes = K.stop_launch_trace(by_shape=True)
        if args.shapes_out and rank == 0:
            with open(args.shapes_out, "w") as f:
                f.write(f"# per (class, shape): launches/step, avg us, % of traced time, TFLOP/s, GB/s (algorithmic)   [{args.steps} serialized steps of {args.frames} frames]\n")
                tot = sum(r["seconds"] for r in shapes.values())
                for (cls, key), r in sorted(shapes.items(), key=lambda kv: -kv[1]["seconds"]):
                    f.write(f"{100 * r['seconds'] / tot:5.2f}%  {r['launches'] // args.steps:4d}x {r['seconds'] / r['launches'] * 1e6:8.1f} us  "
                            f"{r['flops'] / r['seconds'] / 1e12:7.1f} TF  {r['bytes'] / r['seconds'] / 1e9:7.0f} GB/s  [{cls}] {key}\n")
        fx.serial = False
        classes, traced_share = roofline_classes(trace, dt_serial)
        for c in classes:
            c["launches_per_step"] = c.pop("launches") // max(args.steps, 1)
        log(f"roofline pass: {args.steps} serialized steps in {dt_serial:.3f}s")

    per_config = None if args.no_per_config else per_config_rates(fx, dev, rank, world, log, dist)

    if rank == 0:
        job_frames = args.frames * args.steps * (1 if strong else world)
        workload = ("BASELINE cfg#5, dense schedule: fused per-frame path on a synthetic 5 s @ 30 fps 1080p clip (150 unique BGR "
                    "frames resident in HBM): YOLOv8-l detect (letterbox 384x640, NMS, scale_boxes; f16 plan) -> SAM (Hiera-B+ image "
                    "encoder + FPN at 1024x1024, box-prompted mask decoder, 1080p mask + statistics + contour features) + DINOv3 "
                    "ViT-L/16 embed (224x224) on EVERY frame; per-frame records packed, gathered to rank 0 and copied to pinned "
                    "host memory inside the step; synthetic weights")
        line = {
            "metric": "frames/sec (whole node) for YOLO+SAM3+DINOv3 feature extraction, 1080p clips",
            "value": job_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": workload,
                       "frames_per_step": args.frames if strong else args.frames * world,
                       "frames_per_gpu_per_step": (hi - lo) if strong else args.frames,
                       "parallelism": (f"ONE clip per step sharded over {world} GPUs in contiguous blocks of {ldist.shard_rows(args.frames, world)} "
                                       f"frames, one gather of the packed records to rank 0 per step (SAM passes of {shard_chunk} frames)"
                                       if strong else f"one clip per GPU, {world} GPU(s), one gather per step"),
                       "gflop_per_frame": sum(GFLOP_PER_FRAME.values()), "streams": fx.max_streams},
        }
        if world > 1:
            line["weak_scaling"] = {"value": args.frames * world * args.steps / dt_dense, "unit": "frames/s", "ms_per_step": dt_dense / args.steps * 1e3,
                                    "note": "every rank its own 150-frame clip per step (per-GPU work fixed), one all-gather of the records per step"}
            line["sharded_clip"] = {"value": args.frames * args.steps / dt_shard, "unit": "frames/s", "ms_per_step": dt_shard / args.steps * 1e3,
                                    "frames_per_gpu_per_step": ldist.shard_rows(args.frames, world), "sam_chunk": shard_chunk,
                                    "note": "north_star's path: decoded frames of ONE clip shard across the GPUs, RCCL gather only to reassemble per-clip outputs"}
        if dt_ref is not None:
            line["reference_schedule"] = {
                "value": args.frames * world * args.steps / dt_ref, "unit": "clip frames/s", "ms_per_clip": dt_ref / args.steps * 1e3,
                "network_passes_per_clip": {"yolo+sam": len(det_idx), "dino": len(emb_idx)}, "yolo_plan": "exact",
                "note": "the services' own sampling (yolo main.py:67, dinov3 main.py:127): YOLO+SAM on frames 0,15,..,135, DINO on 0,30,..,120 "
                        "of the same clip (one clip per GPU); the mode whose JSON matches the reference: YOLO on its exact plan "
                        "(keep-sets identical to the fp32 path)"}
        if classes:
            modelled = [c for c in classes if c["bound"]]
            head = max(modelled, key=lambda c: c["time_share"])
            line["roofline"] = dict(head, measured_on=f"the dense clip-per-GPU steps replayed on one HIP stream after the timed region "
                                                      f"({dt_serial / args.steps * 1e3:.1f} ms/step serialized; events cover "
                                                      f"{100 * traced_share:.0f} % of it)")
            line["roofline_classes"] = classes
        if per_config:
            line["per_config"] = per_config
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames, 100)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
