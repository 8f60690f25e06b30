"""Development aid: run one FusedExtractor step with every tensor the host side allocates (torch.empty / zeros / *_like /
new_empty / new_zeros on the GPU) wrapped in 64 KB guard bands filled with a sentinel, then report every allocation
whose guard bands were written — i.e. a kernel that stores outside the buffer it was given.  There is no GPU address
sanitizer on this pool; this is the poor man's one.

  python tools/guard_probe.py [n_frames]"""
import math
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]

G = 65536
SENT = 0xA5
records = []
_empty = torch.empty


def _is_cuda(device):
    return device is not None and torch.device(device).type == "cuda"


def guarded_empty(*size, dtype=None, device=None, **kw):
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        size = tuple(size[0])
    if not _is_cuda(device) or kw:
        return _empty(*size, dtype=dtype, device=device, **kw)
    dtype = dtype or torch.float32
    nbytes = math.prod(size) * _empty((), dtype=dtype).element_size()
    if nbytes == 0:
        return _empty(*size, dtype=dtype, device=device)
    pad = (-nbytes) % 16
    base = _empty(nbytes + pad + 2 * G, dtype=torch.uint8, device=device)
    base[:G].fill_(SENT)
    base[G + nbytes:].fill_(SENT)
    records.append((base, nbytes, "".join(traceback.format_stack(limit=5)[:-1])))
    return base[G:G + nbytes].view(dtype).view(size)


def guarded_zeros(*size, dtype=None, device=None, **kw):
    if not _is_cuda(device) or kw:
        return _zeros(*size, dtype=dtype, device=device, **kw)
    return guarded_empty(*size, dtype=dtype, device=device).zero_()


_zeros = torch.zeros
_empty_like, _zeros_like = torch.empty_like, torch.zeros_like
torch.empty = guarded_empty
torch.zeros = guarded_zeros
torch.empty_like = lambda t, **kw: guarded_empty(t.shape, dtype=kw.get("dtype", t.dtype), device=t.device) if t.is_cuda else _empty_like(t, **kw)
torch.zeros_like = lambda t, **kw: guarded_empty(t.shape, dtype=kw.get("dtype", t.dtype), device=t.device).zero_() if t.is_cuda else _zeros_like(t, **kw)
torch.Tensor.new_empty = lambda t, *size, **kw: guarded_empty(*size, dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device))
torch.Tensor.new_zeros = lambda t, *size, **kw: guarded_empty(*size, dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device)).zero_()

from lmx import pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(n)], 0)).to(dev)
for serial in (True, False):
    records.clear()
    fx.serial = serial
    out = fx.step(frames, keep_byte_masks=True)
    torch.cuda.synchronize()
    bad = 0
    sites = {}
    for base, nbytes, site in records:
        head = base[:G] != SENT
        tail = base[G + nbytes:] != SENT
        nh, nt = int(head.sum()), int(tail.sum())
        if nh or nt:
            bad += 1
            key = site
            if key not in sites:
                first_t = int(tail.nonzero()[0]) if nt else -1
                last_h = G - int(head.nonzero()[-1]) if nh else -1
                sites[key] = [0, nbytes, nh, nt, last_h, first_t]
            sites[key][0] += 1
    print(f"serial={serial}: {len(records)} guarded allocations, {bad} with written guard bands", flush=True)
    for site, (cnt, nbytes, nh, nt, last_h, first_t) in sites.items():
        print(f"--- {cnt} allocation(s), e.g. {nbytes} bytes: {nh} guard bytes written before (reaching {last_h} B before the start), "
              f"{nt} after (first at +{first_t} B past the end)\n{site}", flush=True)
