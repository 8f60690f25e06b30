"""HIP graphs for single-frame calls.  ONE frame through a model is 200 - 700 C-ABI launches of a few microseconds of GPU work each
— the shape of call the reference's per-frame loops make through lmx.adapters (yolo main.py:76, sam3 main.py:80-88, dinov3
main.py:107-113).  `GraphedFn` captures such a call once per input signature into a HIP graph (torch.cuda.CUDAGraph records the
library's launches on the capture stream: they go to torch's current stream of the operands' device, lmx/kernels.py) and replays
it afterwards: one graph launch per call, identical kernels, identical bits (tests/test_gpu_adapters.py).

MEASURED (tools/latency_probe.py, profiles/r03_latency_probe.txt): the replay takes as long as the eager call — YOLOv8-l on a
1080p frame 4.02 ms against 4.01, SAM Hiera-B+ set_image + predict 4.43 / 4.42, SAM ViT-B 5.88 / 5.86, DINOv3-L 4.76 / 4.72 — a
frame's latency is the GPU's: a chain of ~600 dependent kernels of a few microseconds each, not the host's launch rate.  What a
replay does save is the host thread (it issues one launch instead of ~600 ctypes calls).  OFF by default for that reason;
LMX_GRAPHS=1 turns it on for the adapters.  Batched calls (the fused service: 10 - 150 frames per launch plan) are bound by their
kernels (profiles/r03_reference_schedule_probe.txt) and always run eagerly."""
import os

import torch


def enabled():
    return os.environ.get("LMX_GRAPHS", "0") == "1"


def _map(out, f):
    if isinstance(out, torch.Tensor):
        return f(out)
    if isinstance(out, dict):
        return {k: _map(v, f) for k, v in out.items()}
    if isinstance(out, (list, tuple)):
        return type(out)(_map(v, f) for v in out)
    return out


class GraphedFn:
    """fn(*tensors) -> (nested) tensors, captured per input signature after `warmup` eager calls on static copies of the inputs.
    The call copies the inputs into the static buffers, replays, and returns CLONES of the static outputs (`clone_outputs=False`:
    the static outputs themselves, valid until the next call — for results the caller consumes at once).
    `extra_key`: anything hashable that changes the captured launches (thresholds passed as kernel arguments, plans).
    A graph pins its workspace, so at most `max_entries` signatures are captured; further ones run eagerly."""

    def __init__(self, fn, warmup=2, clone_outputs=True, max_entries=4):
        self.fn, self.warmup, self.clone, self.max_entries = fn, warmup, clone_outputs, max_entries
        self.cache = {}
        self.failed = False  # a capture that raised once is not retried: the call runs eagerly from then on

    def __call__(self, *inputs, extra_key=None):
        if self.failed or not enabled() or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in inputs):
            return self.fn(*inputs)
        key = (extra_key,) + tuple((tuple(t.shape), t.dtype, t.device.index) for t in inputs)
        ent = self.cache.get(key)
        if ent is None and len(self.cache) >= self.max_entries:
            return self.fn(*inputs)
        if ent is None:
            statics = [torch.empty_like(t) for t in inputs]
            for s, t in zip(statics, inputs):
                s.copy_(t)
            try:
                for _ in range(self.warmup):  # first-use work (table uploads, weight splits, kernel attributes) happens eagerly
                    self.fn(*statics)
                torch.cuda.synchronize(inputs[0].device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self.fn(*statics)
            except Exception as e:  # noqa: BLE001 — a call that cannot be captured still has to be served
                self.failed = True
                print(f"lmx.graphs: capture failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize(inputs[0].device)
                return self.fn(*inputs)
            ent = self.cache[key] = (g, statics, out)
        g, statics, out = ent
        for s, t in zip(statics, inputs):
            s.copy_(t)
        g.replay()
        return _map(out, lambda t: t.clone()) if self.clone else out
