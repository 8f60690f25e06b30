"""The reference schedule (YOLO + SAM on frames 0, 15, .., 135, DINO on 0, 30, .., 120 of a 150-frame clip; exact plans) on one GPU:
wall time per clip with the step's streams and on one stream, C-ABI launches per clip, and (under rocprofv3 --kernel-trace --stats)
the sum of kernel durations per clip — is the mode bound by launches or by kernels?  Usage: python tools/ref_schedule_probe.py [n]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 30  # frames per SAM pass
dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
frames = torch.from_numpy(synth.synth_clip(100, 150)).to(dev)
sched = sorted(set(range(0, 150, 15)) | set(range(0, 150, 30)))
sf = frames[sched].contiguous()
det = [j for j, i in enumerate(sched) if i % 15 == 0]
emb = [j for j, i in enumerate(sched) if i % 30 == 0]


def run(k, prec):
    for _ in range(3):
        fx.step(sf, sam_chunk=CH, det_idx=det, emb_idx=emb, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fx.step(sf, sam_chunk=CH, det_idx=det, emb_idx=emb, precision=prec)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for prec in ("exact", "f16"):
    fx.serial = False
    ms = run(n, prec)
    fx.serial = True
    ms1 = run(n, prec)
    K.start_launch_trace()
    fx.step(sf, sam_chunk=CH, det_idx=det, emb_idx=emb, precision=prec)
    tr = K.stop_launch_trace()
    nl = sum(r["launches"] for r in tr.values())
    ev = sum(r["seconds"] for r in tr.values()) * 1e3
    print(f"reference schedule (SAM passes of {CH}), plans {prec:5s}: {ms:6.2f} ms per clip on the step's streams, {ms1:6.2f} ms on one stream; {nl} C-ABI launches per clip, "
          f"event-timed kernel time {ev:6.2f} ms", flush=True)

if os.environ.get("LMX_REF_SHAPES"):  # per-shape table of one exact-plan clip (serialized, event-timed)
    fx.serial = True
    K.start_launch_trace()
    for _ in range(3):
        fx.step(sf, sam_chunk=CH, det_idx=det, emb_idx=emb, precision="exact")
    _, shapes = K.stop_launch_trace(by_shape=True)
    tot = sum(r["seconds"] for r in shapes.values())
    print(f"# reference schedule, exact plans: {tot / 3 * 1e3:.2f} ms per clip (event-timed, one stream); top shapes:")
    for (cls, key), r in sorted(shapes.items(), key=lambda kv: -kv[1]["seconds"])[:30]:
        print(f"{100 * r['seconds'] / tot:5.1f}%  {r['launches'] // 3:3d}x {r['seconds'] / r['launches'] * 1e6:7.1f} us  {r['flops'] / max(r['seconds'], 1e-12) / 1e12:6.0f} TF  {r['bytes'] / max(r['seconds'], 1e-12) / 1e9:6.0f} GB/s  [{cls}] {key}")
