"""Development aid (needs a -DLMX_DBG_TIMELINE build: make O=obj_tl EXTRA=-DLMX_DBG_TIMELINE LIB=../lmx/liblmx_tl.so): per-tile
timeline of one GEMM launch — which CU ran the tile, when it started, when its main loop ended, when its epilogue ended —
to see whether co-resident workgroups overlap one's epilogue with the other's main loop.
  LMX_LIB=.../liblmx_tl.so python tools/gemm_timeline_probe.py C"""
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
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.lmx_dbg_get_timeline.argtypes = [C.c_void_p, C.c_int64]
g = torch.Generator(device=dev).manual_seed(0)
M, N, Kd, f32, act, res = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "122880,1792,448,0,2,0".split(","))]
a = torch.randn(M, Kd, device=dev, generator=g).half()
w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).half()
dt = torch.float32 if f32 else torch.float16
o = torch.empty(M, N, device=dev, dtype=dt)
r = torch.randn(M, N, device=dev, generator=g).to(dt) if res else None
b = torch.randn(N, device=dev, generator=g)
for v in sys.argv[1].split(","):
    var = v
    lib.lmx_dbg_set_gemm2_variant(0 if var == "default" else ord(var))
    for _ in range(3):
        K.gemm(a, w, bias=b, act=act, res=r, out=o)
    torch.cuda.synchronize()
    tl = np.zeros((16384, 8), dtype=np.uint64)
    nt_ = ((M + 255) // 256) * ((N + (127 if var in "CDASTUV" else 255)) // (128 if var in "CDASTUV" else 256))
    raw.lmx_dbg_get_timeline(tl.ctypes.data_as(C.c_void_p), tl.nbytes)
    tl = tl[:nt_]
    t0 = tl[:, 2].min()
    hw, xcc = tl[:, 0].astype(np.int64), tl[:, 1].astype(np.int64) & 0xf
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    st, me, en = [(tl[:, i].astype(np.int64) - int(t0)) / 100.0 for i in (2, 3, 4)]  # us
    si, dr = [(tl[:, i].astype(np.int64) - int(t0)) / 100.0 for i in (6, 7)]
    print(f"  tile start -> first k-tile landed and all waves there {np.mean(si - st):.2f} us, -> second k-tile {np.mean(dr - si):.2f} us "
          f"(k-loop {np.mean(me - st):.2f} us)")
    print(f"variant {v}: {len(tl)} tiles on {len(np.unique(cu))} CUs, span {en.max():.1f} us; main loop {np.mean(me - st):.2f} us, "
          f"epilogue {np.mean(en - me):.2f} us per tile")
    # co-residency: at each tile's epilogue midpoint, is another tile of the same CU in its main loop / epilogue?
    both_epi = one_main = alone = 0
    for c in np.unique(cu):
        ix = np.where(cu == c)[0]
        for i in ix:
            mid = 0.5 * (me[i] + en[i])
            others = [j for j in ix if j != i and st[j] <= mid < en[j]]
            if not others:
                alone += 1
            elif any(mid >= me[j] for j in others):
                both_epi += 1
            else:
                one_main += 1
    n = len(tl)
    print(f"  at a tile's epilogue midpoint the CU's other resident tile is: in its main loop {100 * one_main / n:.0f} %, also in its "
          f"epilogue {100 * both_epi / n:.0f} %, absent {100 * alone / n:.0f} %")
    c = np.unique(cu)[5]
    ix = np.where(cu == c)[0]
    ix = ix[np.argsort(st[ix])][:10]
    print("  one CU, first tiles: " + "  ".join(f"[{st[i]:.1f} {me[i]:.1f} {en[i]:.1f} b{int(tl[i, 5])}]" for i in ix))
    firsts = np.sort(st)[:600]
    print(f"  start times of the first 600 tiles: 256th {firsts[255]:.1f} us, 512th {firsts[511]:.1f} us, 600th {firsts[599]:.1f} us")
