"""Where does the f16 path's deviation from the fp32 YOLO oracle come from, and what would a higher-precision part buy?
  CPU experiment on the oracle (YOLOv8-n, one synthetic 1080p frame, conf 0.25): the f16 STORAGE
emulation of oracle.yolo is applied to a subset of the 119 rounding points (weights and stored activations, in network
order) and the result is pushed through the keep-set margin rule (tests/keepset.py).  Output columns:
eps_score, eps_iou, firm, ambiguous, |keep-set difference|, fp32 keep-set size.   Result (profiles/r02_yolo_precision_probe.txt):
the FIRST quarter of the network (stem + first two stages, the high-resolution layers) alone reproduces the full
deviation (3.8e-3); the last quarter (PAN tail + Detect head) contributes 6e-4; rounding only the weights gives 2.2e-3,
only the activations 3.1e-3.  A higher-precision Detect head or last C2f stage therefore buys nothing measurable; shrinking
eps by 10x needs split-f16 (hi + lo) operands for BOTH weights and activations through the whole backbone, i.e. three MFMA
passes per GEMM = 3x YOLO's 11 % of the FLOPs (about -18 % frames/s).  Not adopted: the margin rule with the measured
eps (3e-3 .. 6e-3) is the parity statement for the f16 path."""
import sys, os, numpy as np, torch
sys.path[:0]=['/root/repo','/root/repo/vision-sam3-yolo-lameless_amd','/root/repo/tests']
from lmx import yolo, synth
from oracle import yolo as OY
import keepset as KS
scale='n'
cfg=yolo.YoloConfig(scale); sd=yolo.synthetic_state_dict(cfg,7,yolo.bn_stats_path(scale))
fr=synth.synth_frame(3,40)
ref=OY.predict(scale,80,sd,fr,conf=0.25)
cref=KS.compact_pred(ref['pred'])
orig_q=OY._q
def run(mode):
    st={'n':0}
    def q(x):
        st['n']+=1
        is_w = x.dim()==4 and x.shape[2] in (1,3) and x.requires_grad is False and x.shape[0] < 2000 and st.get('in_conv',False)
        return orig_q(x) if mode(st['n'], x) else x
    OY._q=q
    OY._EMULATE_F16=True
    try:
        r=OY.predict(scale,80,sd,fr,conf=0.25,emulate_f16=True)
    finally:
        OY._q=orig_q
    c=KS.compact_pred(r['pred'])
    es,ei,_=KS.measure_eps(cref,c,0.25)
    firm,amb,_=KS.classify(cref,0.25,0.7,es,ei)
    a,b=set(r['src'].tolist()),set(ref['src'].tolist())
    return st['n'], es, ei, len(firm), len(amb), len(a^b), len(b)
n,*r=run(lambda i,x: True); print('all', n, r)
N=n
print('none', run(lambda i,x: False)[1:])
for lo,hi in [(0,N//4),(N//4,N//2),(N//2,3*N//4),(3*N//4,N)]:
    print('only calls',lo,hi, run(lambda i,x,lo=lo,hi=hi: lo<i<=hi)[1:])
# weights only vs activations only: weights are 4-D tensors with small spatial dims (k x k); activations have large H,W
print('weights only', run(lambda i,x: x.dim()==4 and x.shape[-1]<=3)[1:])
print('acts only', run(lambda i,x: not (x.dim()==4 and x.shape[-1]<=3))[1:])
