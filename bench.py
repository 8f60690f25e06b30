#!/usr/bin/env python3
"""bench.py — frames/sec of the fused YOLO + SAM(Hiera-B+) + DINOv3 feature-extraction path on synthetic 1080p clips
(BASELINE.json metric).  One process per GPU; a step = one pass of the hot path over one batch of `--frames` 1080p
frames already resident in HBM (dense schedule: every frame through all three networks).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the dominant kernel = the MFMA GEMM,
timed live with HIP events on the launch stream) and `cpu_baseline` (the fp32 oracle on the host cores, bounded sample).
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
# algorithmic FLOPs per 1080p frame (SURVEY.md §8d cfg#5): YOLOv8-l @384x640 + Hiera-B+ trunk+FPN @1024^2 + DINOv3 ViT-L/16 @224^2
GFLOP_PER_FRAME = {"yolo": 99.1, "sam": 645.0, "dino": 125.7}


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
    bn = os.path.join(ROOT, "tests", "golden", "yolov8l_bn_w7.npz")
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
            box = np.array([[300.0, 150.0, 1200.0, 900.0]], np.float32)
            sp = OD.prompt_encode_box(msd, torch.from_numpy(OD.scale_box(box, f.shape[:2], (576, 1024))))
            low, _ = OD.mask_decode(msd, fpn[2], sp)
            OD.postprocess(low, (576, 1024), f.shape[:2])
            OV.embed(dcfg, dsd, torch.from_numpy(OP.dino_pixel_values(f))[None])
    dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_frames} synthetic 1080p frames, YOLOv8-l + Hiera-B+ encoder + SAM mask decoder + DINOv3 ViT-L/16, fp32 PyTorch CPU, batch 1"}


def pmc_gemm_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 --pmc passes of this same command (tools/pmc.sh ->
    tools/pmc_summary.py -> profiles/r01_pmc_summary.json; FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE),
    launch-weighted over the GEMM kernels.  Counters cannot be read inside the timed run, so this is the profiled figure."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rows = json.load(f)
    n = b = 0.0
    for name, r in rows.items():
        if "gemm2_kernel" in name or "gemm_kernel" in name:
            n += r["launches"]
            b += r["launches"] * (r["read_MB"] + r["write_MB"]) * 1e6
    return b / n if n else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64, help="1080p frames per GPU per step")
    ap.add_argument("--sam-chunk", type=int, default=16, help="frames per SAM encoder pass (each pass runs on its own HIP stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=4)
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
    log("generating synthetic frames")
    # each rank owns a contiguous block of the synthetic clip (weak scaling: per-GPU work fixed)
    host = np.stack([synth.synth_frame(100 + rank, i) for i in range(min(args.frames, 8))], 0)
    host = np.concatenate([host] * (-(-args.frames // host.shape[0])), 0)[:args.frames]
    frames = torch.from_numpy(host).to(dev)

    def step():
        out = fx.step(frames, sam_chunk=args.sam_chunk)
        return ldist.gather_frame_records(out) if world > 1 else out

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i + 1}/{args.warmup} done")
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
    log(f"timed region: {args.steps} steps in {dt:.3f}s")
    # Roofline leg: the timed steps keep three HIP streams in flight (YOLO+DINO beside two SAM passes), so an event pair
    # around one launch would time its neighbours too.  The GEMM launches are therefore bracketed on K more steps of the
    # SAME workload run on one stream (not part of `value`); rocprofv3's per-kernel averages in profiles/ are taken the
    # same way (LMX_SERIAL=1).
    # (every rank replays, without collectives, so that all ranks reach the closing all-reduce together; rank 0 reports)
    fx.serial = True
    fx.step(frames, sam_chunk=args.sam_chunk)
    torch.cuda.synchronize()
    K.start_gemm_trace()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        fx.step(frames, sam_chunk=args.sam_chunk)
    torch.cuda.synchronize()
    dt_serial = time.perf_counter() - t1
    g_flops, g_secs, g_launches = K.stop_gemm_trace()
    fx.serial = False
    log(f"roofline pass: {args.steps} serialized steps in {dt_serial:.3f}s")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_frames = args.frames * world * args.steps
        achieved = g_flops / g_secs / 1e12 if g_secs > 0 else 0.0
        line = {
            "metric": "frames/sec (whole node) for YOLO+SAM3+DINOv3 feature extraction, 1080p clips",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "fused dense per-frame path: YOLOv8-l detect (letterbox 384x640, NMS, scale_boxes) -> SAM "
                                   "(Hiera-B+ image encoder + FPN at 1024x1024, box-prompted mask decoder, 1080p mask + stats) "
                                   "+ DINOv3 ViT-L/16 embed (224x224) on every frame of synthetic 1080p BGR clips resident in "
                                   "HBM; synthetic weights",
                       "frames_per_gpu_per_step": args.frames, "parallelism": f"frames sharded over {world} GPU(s)",
                       "gflop_per_frame": sum(GFLOP_PER_FRAME.values())},
            "roofline": {"bound": "mfma", "kernel": "gemm2_kernel / gemm_kernel (lmx_k_gemm: every Linear, 1x1 and 3x3-implicit-GEMM launch)",
                         "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F16_TFLOPS,
                         "traffic": pmc_gemm_traffic(), "launches_per_step": g_launches // max(args.steps, 1),
                         "flop_per_launch": g_flops / max(g_launches, 1),
                         "gemm_time_share": g_secs / dt_serial if dt_serial > 0 else None,
                         "measured_on": "the same steps replayed on one HIP stream after the timed region "
                                        f"({dt_serial / args.steps * 1e3:.1f} ms/step serialized)"},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames, 100)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
