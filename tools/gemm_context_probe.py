"""Development aid: why does the qkv GEMM of Hiera stage 3 (M=122880, N=1344, K=448) take 234 us inside the step and 175 us in a
loop of its own?  Times it (a) alone on one buffer set, (b) rotating over buffer sets larger than the Infinity Cache, (c) each
launch preceded by the LayerNorm that produces its input, as in the model."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M, N, Kd = 122880, 1344, 448
w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).half()
b = torch.randn(N, device=dev, generator=g)
gam, bet = torch.ones(Kd, device=dev), torch.zeros(Kd, device=dev)
NS = 4
xs = [torch.randn(M, Kd, device=dev, generator=g) for _ in range(NS)]
hs = [x.half() for x in xs]
outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(NS)]


def timed(fn, n=12):
    fn(0)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        fn(i, ev[i])
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b_) * 1000 for a, b_ in ev)
    return ts[len(ts) // 2], ts[0]


def gemm_only(i, ev=None, rot=False):
    k = i % NS if rot else 0
    if ev:
        ev[0].record()
    K.gemm(hs[k], w, bias=b, out=outs[k])
    if ev:
        ev[1].record()


def ln_then_gemm(i, ev=None):
    k = i % NS
    K.layernorm(xs[k], gam, bet, 1e-6, out=hs[k])
    if ev:
        ev[0].record()
    K.gemm(hs[k], w, bias=b, out=outs[k])
    if ev:
        ev[1].record()


def ln_rev_then_gemm(i, ev=None, parts=4):
    """the producer writes its LAST rows first (emulating opposite traversal directions of producer and consumer): the rows the
    GEMM reads first are then the most recently written ones"""
    k = i % NS
    step = M // parts
    for q_ in reversed(range(parts)):
        K.layernorm(xs[k][q_ * step:(q_ + 1) * step], gam, bet, 1e-6, out=hs[k][q_ * step:(q_ + 1) * step])
    if ev:
        ev[0].record()
    K.gemm(hs[k], w, bias=b, out=outs[k])
    if ev:
        ev[1].record()


from lmx import _lib  # noqa: E402

lib = _lib.load()
for var in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["default"]):
  lib.lmx_dbg_set_gemm2_variant(0 if var == "default" else ord(var))
  print("variant", var)
  print("one buffer set           median %.1f us, min %.1f us" % timed(lambda i, ev=None: gemm_only(i, ev, False)))
  print("rotating 4 buffer sets   median %.1f us, min %.1f us" % timed(lambda i, ev=None: gemm_only(i, ev, True)))
  print("LayerNorm then GEMM      median %.1f us, min %.1f us (the GEMM alone, timed between events)" % timed(ln_then_gemm))
  print("LayerNorm in 4 parts, last part first, then GEMM   median %.1f us, min %.1f us" % timed(ln_rev_then_gemm))
  print("LayerNorm in 16 parts, last part first, then GEMM  median %.1f us, min %.1f us" % timed(lambda i, ev=None: ln_rev_then_gemm(i, ev, 16)))
