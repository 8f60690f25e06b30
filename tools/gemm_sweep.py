"""Development aid: time every distinct GEMM shape of one fused step under the gemm2 tiling selected by LMX_GEMM2_VARIANT
(one process per variant: the launcher reads the variable once), to check the launcher's per-shape choice.
  python tools/gemm_sweep.py out.json            (run once per variant; compare the JSON files offline: tools/gemm_sweep_report.py)"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib, pipeline, synth  # noqa: E402
from lmx import kernels as K  # noqa: E402
from lmx._lib import GemmDesc  # noqa: E402

dev = torch.device("cuda:0")
fx = pipeline.FusedExtractor(dev)
fx.serial = True
NF = int(os.environ.get("LMX_SWEEP_FRAMES", "16"))  # 30 = the bench's SAM pass (--sam-chunk 30)
frames = torch.from_numpy(synth.synth_clip(100, NF)).to(dev)
fx.step(frames, sam_chunk=NF)
torch.cuda.synchronize()
lib = _lib.load()
raw = lib.lmx_k_gemm
seen = {}


def spy(desc_ref, stream):
    d = desc_ref._obj
    if d.a_mode == 0 and d.M >= 512 and d.N >= 96:  # the gemm2 family (plain GEMMs; convolutions have two candidates only)
        key = (d.M, d.N, d.K, d.out_dtype, d.act, 1 if d.res else 0, 1 if d.bias else 0, 1 if d.scale else 0, d.res_rows)
        seen[key] = seen.get(key, 0) + 1
    return raw(desc_ref, stream)


class P:
    def __getattr__(self, name):
        return spy if name == "lmx_k_gemm" else getattr(lib, name)


_lib._lib = P()
fx.step(frames, sam_chunk=NF)
torch.cuda.synchronize()
_lib._lib = lib
VARIANTS = ["default", "C", "D", "E", "Y", "Z"]
# LMX_SWEEP_COLD=1: rotate over buffer sets larger than the 256 MB Infinity Cache between launches.  A loop over ONE set keeps
# the A operand cache-resident, which flatters the small tilings (they re-read A more often): the qkv GEMM of Hiera stage 3
# measures 176 us that way and 235 us behind the LayerNorm that produces its input (tools/gemm_context_probe.py)
COLD = bool(os.environ.get("LMX_SWEEP_COLD"))
ROUNDS = 3
out = {}
g = torch.Generator(device=dev).manual_seed(0)
for key, cnt in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
    M, N, Kd, od, act, res, bias, scale, rr = key
    dt = torch.float32 if od == 1 else torch.float16
    # (the A operands alone must exceed the cache: the large f16 outputs are written with non-temporal stores and do not displace them)
    NS = max(2, min(8, -(-600_000_000 // (M * Kd * 2)))) if COLD else 1
    a_s = [torch.randn(M, Kd, device=dev, generator=g).half() for _ in range(NS)]
    w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).half()
    o_s = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(NS)]
    r_s = [torch.randn(rr or M, N, device=dev, generator=g).to(dt) if res else None for _ in range(NS if not rr else 1)]
    a, o, r = a_s[0], o_s[0], r_s[0]
    b = torch.randn(N, device=dev, generator=g) if bias else None
    s = torch.rand(N, device=dev, generator=g) if scale else None
    best = {v: 1e30 for v in VARIANTS}
    for rnd in range(ROUNDS):  # interleaved rounds in ONE process (guide rule 24): minimum per variant
        for v in VARIANTS:
            lib.lmx_dbg_set_gemm2_variant(0 if v == "default" else ord(v))
            K.gemm(a, w, bias=b, act=act, scale=s, res=r, out=o, res_rows=rr)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i6 in range(6):
                K.gemm(a_s[i6 % NS], w, bias=b, act=act, scale=s, res=r_s[i6 % len(r_s)], out=o_s[i6 % NS], res_rows=rr)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) * 1000.0 / 6)
    lib.lmx_dbg_set_gemm2_variant(0)
    out["|".join(str(v) for v in key)] = dict(count=cnt, us=best)
    del a, w, o, r, a_s, o_s, r_s
json.dump(out, open(sys.argv[1], "w"), indent=0)
tot_d = sum(v["us"]["default"] * v["count"] for v in out.values())
tot_b = sum(min(v["us"].values()) * v["count"] for v in out.values())
print(f"GEMM time per pass: launcher's choice {tot_d / 1e3:.2f} ms, best tiling per shape {tot_b / 1e3:.2f} ms ({100 * (1 - tot_b / tot_d):.1f} % less)")
for key, v in sorted(out.items(), key=lambda kv: -kv[1]["us"]["default"] * kv[1]["count"])[:40]:
    bn = min(v["us"], key=v["us"].get)
    print(f"{v['us']['default'] * v['count'] / 1e3:6.2f} ms  {key:44s} default {v['us']['default']:7.1f}  best {bn:7s} {v['us'][bn]:7.1f}  " +
          " ".join(f"{n}:{u:.0f}" for n, u in v["us"].items()))
