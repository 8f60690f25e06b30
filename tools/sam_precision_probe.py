"""Which f16 rounding of the SAM v1 ViT-B path carries the mask error: the fp32 oracle with the device
path's f16 STORAGE points switched on per component (oracle/sam_vit.py, oracle/sam_decoder.py EMULATE), mask IoU and relative
low-res logit error against the plain fp32 run.  CPU only (ViT-B at 1024^2: ~20 s per variant and frame).
Usage: python tools/sam_precision_probe.py [frame_slot]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd"), os.path.join(ROOT, "tests", "golden")]
from lmx import sam, sam_decoder, synth, weights  # noqa: E402
from oracle import preprocess as OP  # noqa: E402
from oracle import sam_decoder as OD  # noqa: E402
from oracle import sam_vit as OV  # noqa: E402

slot = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = np.load(os.path.join(ROOT, "tests", "golden", "sam_mask_vit_b_w9.npz"))
seed = int(g["weight_seed"])
frame = synth.synth_frame(int(g["clip_seed"]), int(g["frame_ids"][slot]))
cfg = sam.sam_vit_b()
sd = weights.synth_state_dict(sam.vit_param_spec(cfg), seed)
dsd = sam_decoder.synthetic_state_dict(seed + 100)
pv = torch.from_numpy(OP.sam_pixel_values(frame, 1024))[None]
hw = frame.shape[:2]
rhw = sam.resize_longest_side(hw[0], hw[1], 1024)
sp = OD.prompt_encode_box(dsd, torch.from_numpy(OD.scale_box(g["boxes"][slot:slot + 1], hw, rhw)))
ENC_ALL = {"w", "ln", "qkv", "rel", "p", "ao", "gelu", "neck", "emb"}
DEC_ALL = {"w", "proj", "mlp", "up", "tok"}
_cache = {}


def run(enc, dec):
    key = frozenset(enc)
    if key not in _cache:
        OV.EMULATE = set(enc)
        with torch.no_grad():
            _cache[key] = OV.encoder_forward(cfg, sd, pv)
        OV.EMULATE = set()
    OD.EMULATE = set(dec)
    with torch.no_grad():
        low, _ = OD.mask_decode(dsd, _cache[key], sp)
        m = OD.postprocess(low, rhw, hw).numpy()[0]
    OD.EMULATE = set()
    return low[0], m


t0 = time.time()
low0, m0 = run(set(), set())
print(f"# SAM v1 ViT-B, golden frame slot {slot} (coverage {m0.mean():.4f}); fp32 run {time.time() - t0:.0f}s; "
      f"matches the committed golden mask: {np.array_equal(m0, np.unpackbits(g['mask_bits'][slot], axis=-1)[:, :hw[1]].astype(bool))}")
print("# variant: mask IoU vs fp32, relative low-res logit error")
variants = [("all f16 storage (the round-2 device path)", ENC_ALL, DEC_ALL),
            ("encoder all, decoder fp32", ENC_ALL, set()),
            ("encoder fp32, decoder all", set(), DEC_ALL),
            ("only the f16 embedding hand-over (emb)", {"emb"}, set()),
            ("only the neck's f16 tensors (neck)", {"neck"}, set()),
            ("only encoder weights (w)", {"w"}, set()),
            ("only LayerNorm outputs + qkv + attn out + GELU hidden (ln qkv ao gelu)", {"ln", "qkv", "ao", "gelu"}, set()),
            ("only rel-pos tables + softmax P (rel p)", {"rel", "p"}, set()),
            ("encoder all but neck + emb", ENC_ALL - {"neck", "emb"}, set()),
            ("decoder: only weights", set(), {"w"}),
            ("decoder: only attention projections (proj)", set(), {"proj"}),
            ("decoder: only the upscaler (up)", set(), {"up"}),
            ("decoder: only token MLPs / hypernet (mlp tok)", set(), {"mlp", "tok"}),
            ("encoder all but neck + emb, decoder all but up", ENC_ALL - {"neck", "emb"}, DEC_ALL - {"up"}),
            ("encoder all but neck + emb, decoder only weights", ENC_ALL - {"neck", "emb"}, {"w"})]
for name, enc, dec in variants:
    low, m = run(enc, dec)
    iou = float((m & m0).sum()) / float((m | m0).sum())
    rel = float((low - low0).norm() / low0.norm())
    print(f"{iou:.5f}  {rel:.2e}  {name}", flush=True)
