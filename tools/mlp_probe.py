"""Development aid: fused LN+MLP (csrc/mlp.hip) against the unfused LN -> GEMM -> GEMM launches at the Hiera stage-1/2 shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib  # noqa: E402
if os.environ.get("LMX_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LMX_DBG_LIB"])
from lmx import kernels as K  # noqa: E402
from perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
for rows, D in [(32 * 65536, 112), (32 * 16384, 224)]:
    x = torch.randn((rows, D), device=dev)
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    w1 = (torch.randn((4 * D, D), device=dev) * D ** -0.5).half()
    w2 = (torch.randn((D, 4 * D), device=dev) * (4 * D) ** -0.5).half()
    b1, b2 = torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev)

    def unfused():
        h = K.layernorm(x, g, b, 1e-6)
        u = K.gemm(h, w1, bias=b1, act=K.ACT_GELU)
        K.gemm(u, w2, bias=b2, res=x, out=x)

    t_u = timeit(unfused, iters=5)
    x.normal_()
    t_f = timeit(lambda: K.ln_mlp(x, g, b, w1, b1, w2, b2, 1e-6), iters=5)
    fl = 16.0 * rows * D * D
    print(f"rows={rows} D={D}: unfused {t_u:.3f} ms  fused {t_f:.3f} ms  ({fl / t_f / 1e9:.0f} TFLOP/s, {8.0 * rows * D / t_f / 1e9:.2f} TB/s algorithmic)",
          flush=True)
