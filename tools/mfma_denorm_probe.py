"""Does v_mfma_f32_16x16x32_f16 honour f16 SUBNORMAL operands on gfx950?  (The exact plans split weights as whi + wlo; an
unscaled wlo is subnormal for |w| < 0.12.)  One lmx_k_gemm launch with subnormal W resp. A; expected 64 * 2^-20."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
one = torch.ones((512, 64), dtype=torch.float16, device=dev)
sub = torch.full((128, 64), 2.0 ** -20, dtype=torch.float16, device=dev)
assert float(sub[0, 0]) == 2.0 ** -20
out = K.gemm(one, sub, out_dtype=torch.float32)
print("A normal, W subnormal 2^-20: out[0,0] =", float(out[0, 0]), "expected", 64 * 2.0 ** -20)
out = K.gemm(torch.full((512, 64), 2.0 ** -20, dtype=torch.float16, device=dev), torch.ones((128, 64), dtype=torch.float16, device=dev),
             out_dtype=torch.float32)
print("A subnormal 2^-20, W normal: out[0,0] =", float(out[0, 0]), "expected", 64 * 2.0 ** -20)
big = K.gemm(torch.ones((1024, 64), dtype=torch.float16, device=dev), torch.full((256, 64), 2.0 ** -20, dtype=torch.float16, device=dev),
             out_dtype=torch.float32)
print("LDS-DMA kernel (M=1024, N=256): out[0,0] =", float(big[0, 0]))
