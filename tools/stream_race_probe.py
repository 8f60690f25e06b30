"""Development aid: is the mask a 64-frame multi-stream step produces equal to mask_post recomputed afterwards (everything
idle) from the very logits that step left behind?  DESIGN.md section 6 tells the story.  Round 2: the LMX_DBG_MASK variants
live in the development build only (make -C vision-sam3-yolo-lameless_amd/csrc dbg; prefix the commands below with
LMX_LIB=$PWD/vision-sam3-yolo-lameless_amd/lmx/liblmx_dbg.so); the cause turned out to be the ARITHMETIC hipcc's SLP
vectoriser generated beside plain loads, not the loads (tools/pk_hazard_probe2.hip, tools/defect_round2.sh).

  LMX_DBG_MASK=1 LMX_STREAM_LAYOUT=rr LMX_MAX_STREAMS=6 python tools/stream_race_probe.py   # plain loads: differs
  LMX_DBG_MASK=2 ...                                                                        # + ordering counters
  LMX_STREAM_LAYOUT=rr LMX_MAX_STREAMS=6 python tools/stream_race_probe.py                  # product kernel: equal"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402
from lmx import pipeline, sam, synth  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(64)], 0)).to(dev)
h, w = 1080, 1920
nh, nw = sam.resize_longest_side(h, w, 1024)
log = []
orig = fx.decoder.predict


def spy(emb, boxes, hw, rhw):
    d = orig(emb, boxes, hw, rhw)
    log.append((d["lowres"], d["mask"], d["stats"], kept.pop()))  # the tensors themselves (no copies, no extra launches)
    return d


fx.decoder.predict = spy
kept = []
_lib, _ptr, _stream = K._lib, K._ptr, K._stream


def mask_post_keep(logits, T, nh_, nw_, h_, w_):
    """kernels.mask_post, but the [n,nh,nw] intermediate survives so that it can be compared too"""
    if os.environ.get("SYNC_BEFORE_POST"):
        torch.cuda.current_stream().synchronize()  # host-side wait for everything enqueued on this stream so far
    n, L, _ = logits.shape
    mask = torch.empty((n, h_, w_), dtype=torch.uint8, device=logits.device)
    stats = torch.empty((n, 8), dtype=torch.int64, device=logits.device)
    ws = torch.empty((n, nh_, nw_), dtype=torch.float32, device=logits.device)
    K.check(_lib.load().lmx_k_mask_post(_ptr(logits), n, L, T, nh_, nw_, h_, w_, _ptr(mask), _ptr(stats), _ptr(ws), _stream(logits.device)), "mask_post")
    kept.append(ws)
    return mask, stats


K.mask_post = mask_post_keep
for rep in range(int(os.environ.get("PROBE_STEPS", "5"))):
    log.clear()
    fx.serial = bool(os.environ.get("SER"))
    fx.step(frames)
    torch.cuda.synchronize()
    msg = []
    if os.environ.get("LMX_DBG_MASK") == "2":  # [finished pass-1 workgroups, pass-2 workgroups that started before all of them]
        print("  ordering counters per chunk:", [(int(st_[0, 7]), int(st_[1, 7])) for _, _, st_, _ in log], flush=True)
    for j, (logits, mask, stats, mid) in enumerate(log):
        m2, s2 = mask_post_keep(logits, 1024, nh, nw, h, w)
        mid2 = kept.pop()
        torch.cuda.synchronize()
        if not torch.equal(m2, mask) or not torch.equal(s2[:, :7], stats[:, :7]) or not torch.equal(mid, mid2):
            bad = (m2 != mask).nonzero()
            print(f"  chunk{j}: values present in the pipeline mask {mask.unique().tolist()}; first differing (frame,y,x) "
                  f"{bad[:6].tolist()} ... last {bad[-3:].tolist()}; pipeline bytes there {mask[m2 != mask][:24].tolist()}", flush=True)
            vals = []
            for fb, y, x in bad[:8].tolist():  # the pre-threshold value of those pixels, recomputed in f32 on the host
                def idx(scale, dst, size):
                    src = max(np.float32(scale) * (np.float32(dst) + np.float32(0.5)) - np.float32(0.5), np.float32(0))
                    i0 = int(src)
                    return i0, i0 + (1 if i0 < size - 1 else 0), np.float32(src - np.float32(i0))
                y0, y1, ly = idx(np.float32(nh) / np.float32(h), y, nh)
                x0, x1, lx = idx(np.float32(nw) / np.float32(w), x, nw)
                m = mid[fb].cpu().numpy()
                one = np.float32(1)
                t0 = (one - lx) * m[y0, x0] + lx * m[y0, x1]
                t1 = (one - lx) * m[y1, x0] + lx * m[y1, x1]
                vals.append(f"{float((one - ly) * t0 + ly * t1):.3e} (taps {m[y0, x0]:.3f} {m[y0, x1]:.3f} {m[y1, x0]:.3f} {m[y1, x1]:.3f})")
            print("    pre-threshold values there:", "; ".join(vals), flush=True)
            msg.append(f"chunk{j}: mask {int((m2 != mask).sum())} px, intermediate {int((mid != mid2).sum())} values differ from the recomputation")
    print(f"step {rep}:", "; ".join(msg) if msg else "masks equal their recomputation", flush=True)
