"""Development aid: where does a GEMM tile's time go?  Run once per build (LMX_LIB = the product library, a -DLMX_DBG_NOEPI
build, a -DLMX_DBG_NOMFMA build) and compare: full, no epilogue, loads only; act=0 vs GELU separates the activation's VALU cost.
  LMX_LIB=... python tools/gemm_decomp_probe.py default,C,E,Y"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib  # noqa: E402
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [(122880, 1792, 448, 0, 2, 0), (122880, 1792, 448, 0, 0, 0), (122880, 1344, 448, 0, 0, 0), (122880, 448, 1792, 1, 0, 1),
          (122880, 448, 1792, 0, 0, 0), (122880, 448, 448, 1, 0, 1), (30150, 4096, 1024, 0, 2, 0), (30150, 1024, 4096, 1, 0, 1)]
VARS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["default", "C", "E", "Y"]
if len(sys.argv) > 2:
    SHAPES = [sh for sh in SHAPES if sh[3] == int(sys.argv[2])]
print("lib", os.environ.get("LMX_LIB", "product"))
for M, N, Kd, f32, act, res in SHAPES:
    a = torch.randn(M, Kd, device=dev, generator=g).half()
    w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).half()
    dt = torch.float32 if f32 else torch.float16
    o = torch.empty(M, N, device=dev, dtype=dt)
    r = torch.randn(M, N, device=dev, generator=g).to(dt) if res else None
    b = torch.randn(N, device=dev, generator=g)
    best = {v: 1e30 for v in VARS}
    ref, same = None, {}
    for v in VARS:  # every tiling must produce the same bits (one k-order, one epilogue rounding sequence)
        lib.lmx_dbg_set_gemm2_variant(0 if v == "default" else ord(v))
        o.zero_()
        K.gemm(a, w, bias=b, act=act, res=r, out=o)
        torch.cuda.synchronize()
        if ref is None:
            ref = o.clone()
        same[v] = bool(torch.equal(o, ref))
    for rnd in range(3):
        for v in VARS:
            lib.lmx_dbg_set_gemm2_variant(0 if v == "default" else ord(v))
            K.gemm(a, w, bias=b, act=act, res=r, out=o)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                K.gemm(a, w, bias=b, act=act, res=r, out=o)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) * 1000.0 / 6)
    fl = 2.0 * M * N * Kd
    print(f"M={M} N={N} K={Kd} f32={f32} act={act} res={res}: " + "  ".join(f"{v}:{u:.1f}us/{fl / u / 1e6:.0f}TF" for v, u in best.items()) +
          ("" if all(same.values()) else f"  BITS DIFFER: {[v for v, ok in same.items() if not ok]}"), flush=True)
    lib.lmx_dbg_set_gemm2_variant(0)
    del a, w, o, r
