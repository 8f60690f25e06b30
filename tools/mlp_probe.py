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

# the streamed-image form (csrc/hiera.hip hiera_mlp_kernel, lmx_k_ln_mlp_img) at the same shapes
from lmx import sam  # noqa: E402

for rows, D in [(30 * 65536, 112), (30 * 16384, 224), (10 * 65536, 112), (10 * 16384, 224)]:
    x = torch.randn((rows, D), device=dev)
    g = torch.Generator().manual_seed(3)
    w1 = (torch.randn((4 * D, D), generator=g) * D ** -0.5).half().float()
    w2 = (torch.randn((D, 4 * D), generator=g) * (4 * D) ** -0.5).half().float()
    z4, z1, o1 = torch.zeros(4 * D), torch.zeros(D), torch.ones(D)
    packed = tuple(torch.from_numpy(a).to(dev) for a in sam.pack_ln_mlp(w1.numpy(), z4.numpy(), w2.numpy(), z1.numpy(), o1.numpy(), z1.numpy(),
                                                                        o1.numpy(), z1.numpy()))
    hn = torch.empty((rows, D), dtype=torch.float16, device=dev)
    w1d, w2d = w1.half().to(dev), w2.half().to(dev)
    gd, bd, b1d, b2d = o1.to(dev), z1.to(dev), z4.to(dev), z1.to(dev)
    t_old = timeit(lambda: K.ln_mlp(x, gd, bd, w1d, b1d, w2d, b2d, 1e-6, next_ln=(gd, bd, hn)), iters=5)
    x.normal_()
    t_new = timeit(lambda: K.ln_mlp_img(x, packed, 1e-6, h_next=hn), iters=5)
    print(f"rows={rows} D={D} (with h_next): csrc/mlp.hip {t_old * 1e3:.0f} us, streamed images {t_new * 1e3:.0f} us", flush=True)
