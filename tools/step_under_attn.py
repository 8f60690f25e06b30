"""Development aid for the wrong-load defect: one-stream FusedExtractor steps while three other
streams run the 128-query-tile attention kernel back to back (the only background that made plain loads of mask_post
return wrong values).  Every output field is compared with the step run on an idle GPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import pipeline, synth  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
fx.serial = True
frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(16)], 0)).to(dev)
ref = {k: v.clone() for k, v in fx.step(frames, keep_byte_masks=True).items()}
torch.cuda.synchronize()
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(16 * 4096, 3 * 448, device=dev, generator=g).half()
aos = [torch.empty(16 * 4096, 448, device=dev, dtype=torch.float16) for _ in range(3)]
bgs = [torch.cuda.Stream() for _ in range(3)]
main = torch.cuda.current_stream()
for rep in range(int(os.environ.get("PROBE_STEPS", "6"))):
    for st, ao in zip(bgs, aos):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            for _ in range(int(os.environ.get("BG_CALLS", "300"))):  # ~1.3 ms each alone: the background must outlast the step
                K.attention(qkv[:, :448], qkv[:, 448:896], qkv[:, 896:], ao, 16, 7, 4096, 4096, 64, 0.125)
    out = fx.step(frames, keep_byte_masks=True)
    torch.cuda.synchronize()
    diff = {k: int((out[k] != ref[k]).sum()) for k in ref if not torch.equal(out[k], ref[k])}
    print(f"step {rep}:", f"differs from the idle step: {diff}" if diff else "every output identical to the idle step", flush=True)
