import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import sam, synth, weights
dev = torch.device("cuda:0")
frames = torch.from_numpy(np.stack([synth.synth_frame(100, i) for i in range(8)], 0)).to(dev)
cfg = getattr(sam, "sam_" + (sys.argv[1] if len(sys.argv) > 1 else "vit_b"))()
enc = sam.SamVitEncoder(cfg, weights.synth_state_dict(sam.vit_param_spec(cfg), 9), dev)
for _ in range(3):
    enc.encode(frames)
torch.cuda.synchronize()
