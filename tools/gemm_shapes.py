"""Development aid: per-shape GEMM time table of one fused step (events on the launch stream)."""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib, pipeline, synth  # noqa: E402
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
host = np.stack([synth.synth_frame(100, i) for i in range(4)], 0)
frames = torch.from_numpy(np.concatenate([host] * (n // 4), 0)).to(dev)
fx.step(frames)
torch.cuda.synchronize()
lib = _lib.load()
raw = lib.lmx_k_gemm
log = []


def traced(desc_ref, stream):
    d = desc_ref._obj
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = raw(desc_ref, stream)
    e1.record()
    log.append(((d.M, d.N, d.K, d.a_mode, d.out_dtype, d.act, 1 if d.res else 0), e0, e1))
    return rc


class P:
    def __getattr__(self, name):
        return traced if name == "lmx_k_gemm" else getattr(lib, name)


_lib._lib = P()
for which in ("yolo", "sam", "dino"):
    log.clear()
    if which == "yolo":
        fx.yolo.detect(frames, conf=0.5)
    elif which == "sam":
        for i in range(0, n, 16):
            fx.sam.encode(frames[i:i + 16])
    else:
        fx.dino.embed_frames(frames)
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for key, a, b in log:
        t = a.elapsed_time(b)
        c = agg.setdefault(key, [0, 0.0])
        c[0] += 1
        c[1] += t
    tot = sum(v[1] for v in agg.values())
    fl = sum(2.0 * k[0] * k[1] * k[2] * v[0] for k, v in agg.items())
    print(f"== {which}: {len(log)} gemm launches, {tot:.2f} ms, {fl / tot / 1e9:.0f} TFLOP/s avg")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        M, N, Kk, am, od, act, res = k
        print(f"  M{M:8d} N{N:5d} K{Kk:5d} mode{am} out{'f32' if od else 'f16'} act{act} res{res}  x{v[0]:3d}  {v[1]:8.3f} ms  "
              f"{2.0 * M * N * Kk * v[0] / v[1] / 1e9:7.0f} TF")
