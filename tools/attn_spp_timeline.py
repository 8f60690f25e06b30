"""Where the time of one (window, head) item goes inside attn_spp_kernel (csrc/attn.hip): s_memtime stamps of workgroup 0, per wave and
item, from a -DLMX_DBG_TIMELINE build:
    make -C vision-sam3-yolo-lameless_amd/csrc O=obj_tl EXTRA=-DLMX_DBG_TIMELINE LIB=../lmx/liblmx_tl.so
    LMX_LIB=vision-sam3-yolo-lameless_amd/lmx/liblmx_tl.so python tools/attn_spp_timeline.py
Stamps: 0 loop top, 1 after the barrier, 2 after issuing the next item's LDS-DMA, 3 after S = K Q^T (Q fragment reads + 52 MFMAs),
4 after the softmax, 5 after O = V^T P^T (56 MFMAs), 6 after s_waitcnt vmcnt(0), 7 after the stores."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib  # noqa: E402
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
frames, D, H, hd, G, ws = 30, 448, 8, 56, 64, 14
nW = (-(-G // ws)) ** 2
rows = frames * G * G
qkv = torch.randn(rows, 3 * D, device=dev, generator=g).half()
padkv = torch.randn(3 * D, device=dev, generator=g).half()
out = torch.empty(rows, D, device=dev, dtype=torch.float16)
for _ in range(3):
    K.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, frames * nW, H, ws * ws, ws * ws, hd, hd ** -0.5,
                window=dict(Gh=G, Gw=G, ws=ws, q_stride=1), pad_k=padkv[D:2 * D], pad_v=padkv[2 * D:])
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((64, 8, 8), np.uint64)
fn = lib.lmx_dbg_get_attn_timeline
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int64]
assert fn(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
t = buf.astype(np.int64)
n_items = int((t[:, 0, 0] > 0).sum())
print(f"workgroup 0 processed {n_items} items; cycles (s_memtime ticks), median over items 2..{n_items - 2}")
names = ["wait at the barrier", "issue next item's DMA", "Q reads + S MFMAs", "softmax", "PV MFMAs", "wait vmcnt(0)", "stores"]
sel = slice(2, max(3, n_items - 1))
for w in range(8):
    d = np.diff(t[sel, w, :], axis=1)
    per_item = np.diff(t[sel, w, 0])
    print(f"wave {w}: " + "  ".join(f"{n} {int(np.median(d[:, i]))}" for i, n in enumerate(names)) + f"  | item period {int(np.median(per_item))}")
