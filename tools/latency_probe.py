"""Per-call latency of the models on ONE 1080p frame (what integration A — the adapters inside the reference's per-frame loops — pays):
YOLOv8-l detect(), SAM set_image + predict(box) with the Hiera-B+ and the SAM v1 ViT-B encoder, DINOv3-L embed; both plans."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import dino, sam, sam_decoder, synth, weights, yolo  # noqa: E402

dev = torch.device("cuda:0")
f = torch.from_numpy(synth.synth_frame(3, 40)[None]).to(dev)
box = torch.tensor([[420.0, 360.0, 1010.0, 850.0]], device=dev)


def lat(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ycfg = yolo.YoloConfig("l")
det = yolo.YoloDetector(ycfg, yolo.synthetic_state_dict(ycfg, 7, yolo.bn_stats_path("l")), dev)
dec = sam_decoder.MaskDecoder(sam_decoder.synthetic_state_dict(105), dev)
hcfg = sam.hiera_b_plus()
hier = sam.HieraEncoder(hcfg, weights.synth_state_dict(sam.param_spec(hcfg), 5), dev)
vcfg = sam.sam_vit_b()
vit = sam.SamVitEncoder(vcfg, weights.synth_state_dict(sam.vit_param_spec(vcfg), 9), dev)
dcfg = dino.dinov3_vitl16()
emb = dino.DinoEmbedder(dcfg, weights.synth_state_dict(dino.param_spec(dcfg), 3), dev)
rhw = sam.resize_longest_side(1080, 1920, 1024)


def seg(enc, prec):
    e2 = enc.encode(f, precision=prec)["fpn"][2]
    return dec.predict(e2.reshape(-1, e2.shape[-1]), box, (1080, 1920), rhw, precision=prec)


for prec in ("exact", "f16"):
    print(f"plan {prec:5s}: YOLOv8-l detect {lat(lambda: det.detect(f, conf=0.5, precision=prec)):6.2f} ms | SAM Hiera-B+ set_image+predict "
          f"{lat(lambda: seg(hier, prec)):6.2f} ms | SAM ViT-B set_image+predict {lat(lambda: seg(vit, prec)):6.2f} ms", flush=True)
print(f"DINOv3-L embed (one frame): {lat(lambda: emb.embed_frames(f)):6.2f} ms")

# the same calls replayed from a HIP graph (lmx/graphs.py; what lmx.adapters does for single-frame calls)
from lmx.graphs import GraphedFn  # noqa: E402

os.environ["LMX_GRAPHS"] = "1"

g_det = GraphedFn(lambda fr: det.detect(fr, conf=0.5))
g_h = GraphedFn(lambda fr, b: seg_b(hier, fr, b))
g_v = GraphedFn(lambda fr, b: seg_b(vit, fr, b))
g_d = GraphedFn(lambda fr: emb.embed_frames(fr))


def seg_b(enc, fr, b):
    e2 = enc.encode(fr)["fpn"][2]
    return dec.predict(e2.reshape(-1, e2.shape[-1]), b, (1080, 1920), rhw)


print(f"graphed, plan exact: YOLOv8-l detect {lat(lambda: g_det(f)):6.2f} ms | SAM Hiera-B+ {lat(lambda: g_h(f, box)):6.2f} ms | "
      f"SAM ViT-B {lat(lambda: g_v(f, box)):6.2f} ms | DINOv3-L embed {lat(lambda: g_d(f)):6.2f} ms")
assert not (g_det.failed or g_h.failed or g_v.failed or g_d.failed)
