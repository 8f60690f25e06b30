"""Development aid: time lmx_k_mask_post (16 frames, 256^2 logits -> 1080p masks) under the current LMX_DBG_MASK setting
(variants other than 0 need LMX_LIB=.../liblmx_dbg.so: make -C vision-sam3-yolo-lameless_amd/csrc dbg)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
logits = torch.randn(16, 256, 256, device="cuda", generator=g)
for _ in range(3):
    K.mask_post(logits, 1024, 576, 1024, 1080, 1920)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    K.mask_post(logits, 1024, 576, 1024, 1080, 1920)
e1.record()
torch.cuda.synchronize()
print(f"LMX_DBG_MASK={os.environ.get('LMX_DBG_MASK', '0')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 16-frame mask_post", flush=True)
