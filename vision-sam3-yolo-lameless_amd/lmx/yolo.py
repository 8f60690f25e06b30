"""YOLOv8 detector on liblmx — the model call of services/yolo-pipeline/app/main.py:76
(``self.yolo_model(frame, verbose=False, conf=...)``): LetterBox -> fused Conv-BN-SiLU CSPDarknet/C2f + PAN +
Detect(DFL) -> non_max_suppression -> scale_boxes.  Architecture restated from the public Ultralytics yolov8.yaml /
nn.modules (the package is not installed).

Layout: activations are NHWC f16.  Every C2f owns ONE buffer [n,H,W,(2+nb)*c]: cv1 writes channels [0,2c), each
bottleneck reads the previous c-wide slice and writes the next one (shortcut fused in the GEMM epilogue), cv2 is a
1x1 GEMM over the whole buffer.  torch.cat never happens: producers write straight into channel slices of the
consumer's buffer (SPPF, the PAN concats and the Detect head included).
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import kernels as K
from . import letterbox as LB
from .exact import split_rows_x3  # noqa: F401  (re-exported: tests and tools reach it through lmx.yolo)

SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.0, 1.0, 512),
          "x": (1.0, 1.25, 512)}
BN_EPS = 1e-3
REG_MAX = 16

COCO_NAMES = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
              "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
              "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
              "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
              "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
              "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
              "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard",
              "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors",
              "teddy bear", "hair drier", "toothbrush"]


def _make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


@dataclass
class YoloConfig:
    scale: str = "l"
    nc: int = 80
    imgsz: int = 640
    kpt_shape: tuple = None  # (K, ndim): Pose head (yolov8-pose.yaml) — the tleap-pipeline consumer, tleap main.py:142-163

    def ch(self, c):
        _, width, max_ch = SCALES[self.scale]
        return _make_divisible(min(c, max_ch) * width, 8)

    def depth(self, n):
        d, _, _ = SCALES[self.scale]
        return max(round(n * d), 1)


def layer_table(cfg):
    """The 23 modules of yolov8.yaml with resolved channels: list of dicts (kind, src, c_out, ...)."""
    c = cfg.ch
    d = cfg.depth
    L = [
        dict(kind="conv", src=-1, c2=c(64), k=3, s=2),                  # 0  P1/2
        dict(kind="conv", src=-1, c2=c(128), k=3, s=2),                 # 1  P2/4
        dict(kind="c2f", src=-1, c2=c(128), n=d(3), shortcut=True),     # 2
        dict(kind="conv", src=-1, c2=c(256), k=3, s=2),                 # 3  P3/8
        dict(kind="c2f", src=-1, c2=c(256), n=d(6), shortcut=True),     # 4
        dict(kind="conv", src=-1, c2=c(512), k=3, s=2),                 # 5  P4/16
        dict(kind="c2f", src=-1, c2=c(512), n=d(6), shortcut=True),     # 6
        dict(kind="conv", src=-1, c2=c(1024), k=3, s=2),                # 7  P5/32
        dict(kind="c2f", src=-1, c2=c(1024), n=d(3), shortcut=True),    # 8
        dict(kind="sppf", src=-1, c2=c(1024)),                          # 9
        dict(kind="up", src=-1),                                        # 10
        dict(kind="cat", src=(-1, 6)),                                  # 11
        dict(kind="c2f", src=-1, c2=c(512), n=d(3), shortcut=False),    # 12
        dict(kind="up", src=-1),                                        # 13
        dict(kind="cat", src=(-1, 4)),                                  # 14
        dict(kind="c2f", src=-1, c2=c(256), n=d(3), shortcut=False),    # 15 (P3/8-small)
        dict(kind="conv", src=-1, c2=c(256), k=3, s=2),                 # 16
        dict(kind="cat", src=(-1, 12)),                                 # 17
        dict(kind="c2f", src=-1, c2=c(512), n=d(3), shortcut=False),    # 18 (P4/16-medium)
        dict(kind="conv", src=-1, c2=c(512), k=3, s=2),                 # 19
        dict(kind="cat", src=(-1, 9)),                                  # 20
        dict(kind="c2f", src=-1, c2=c(1024), n=d(3), shortcut=False),   # 21 (P5/32-large)
        dict(kind="detect", src=(15, 18, 21)),                          # 22
    ]
    # resolve input channels
    out_c = []
    for i, m in enumerate(L):
        if m["kind"] in ("conv", "c2f", "sppf"):
            m["c1"] = 3 if i == 0 else out_c[i - 1]
            out_c.append(m["c2"])
        elif m["kind"] == "up":
            out_c.append(out_c[i - 1])
        elif m["kind"] == "cat":
            a, b = m["src"]
            out_c.append(out_c[i - 1] + out_c[b])
        else:
            m["ch"] = tuple(out_c[j] for j in m["src"])
            out_c.append(0)
    return L


def _conv_spec(s, name, c1, c2, k, rms_in=1.0):
    # synthetic init only: the gain compensates the expected RMS of the conv's input (residual sums, concats, pools)
    s[name + ".conv.weight"] = ((c2, c1, k, k), f"wc@{1.68 / rms_in:.4f}")
    s[name + ".bn.weight"] = ((c2,), "g")
    s[name + ".bn.bias"] = ((c2,), "bnb")
    s[name + ".bn.running_mean"] = ((c2,), "b")
    s[name + ".bn.running_var"] = ((c2,), "var")


def param_spec(cfg):
    """Ordered {ultralytics state-dict name: (shape, init kind)} for DetectionModel(yolov8{scale}.yaml)."""
    s = {}
    for i, m in enumerate(layer_table(cfg)):
        p = f"model.{i}"
        if m["kind"] == "conv":
            _conv_spec(s, p, m["c1"], m["c2"], m["k"], 0.45 if i == 0 else 1.0)  # stem input: pixels/255
        elif m["kind"] == "c2f":
            c = m["c2"] // 2
            nb = m["n"]
            # mean square of the chunks [y0a, y0b, y1..yn]: with shortcuts y_j = y_{j-1} + t_j grows like 1 + j
            ms = [1.0, 1.0] + [(1.0 + j if m["shortcut"] else 1.0) for j in range(1, nb + 1)]
            _conv_spec(s, p + ".cv1", m["c1"], 2 * c, 1)
            _conv_spec(s, p + ".cv2", (2 + nb) * c, m["c2"], 1, math.sqrt(sum(ms) / len(ms)))
            for j in range(nb):
                _conv_spec(s, p + f".m.{j}.cv1", c, c, 3, math.sqrt(ms[1 + j]))
                _conv_spec(s, p + f".m.{j}.cv2", c, c, 3)
        elif m["kind"] == "sppf":
            c_ = m["c1"] // 2
            _conv_spec(s, p + ".cv1", m["c1"], c_, 1)
            _conv_spec(s, p + ".cv2", c_ * 4, m["c2"], 1, 1.6)  # chained 5x5 max pools of SiLU outputs
        elif m["kind"] == "detect":
            ch = m["ch"]
            c2 = max(16, ch[0] // 4, REG_MAX * 4)
            c3 = max(ch[0], min(cfg.nc, 100))
            for l, x in enumerate(ch):
                _conv_spec(s, p + f".cv2.{l}.0", x, c2, 3)
                _conv_spec(s, p + f".cv2.{l}.1", c2, c2, 3)
                s[p + f".cv2.{l}.2.weight"] = ((4 * REG_MAX, c2, 1, 1), "w")
                s[p + f".cv2.{l}.2.bias"] = ((4 * REG_MAX,), "boxb")
                _conv_spec(s, p + f".cv3.{l}.0", x, c3, 3)
                _conv_spec(s, p + f".cv3.{l}.1", c3, c3, 3)
                s[p + f".cv3.{l}.2.weight"] = ((cfg.nc, c3, 1, 1), "w")
                s[p + f".cv3.{l}.2.bias"] = ((cfg.nc,), "clsb")
            if cfg.kpt_shape is not None:  # Pose: cv4 = Conv(x,c4,3) -> Conv(c4,c4,3) -> Conv2d(c4,nk,1), c4 = max(ch[0]//4, nk)
                nk = cfg.kpt_shape[0] * cfg.kpt_shape[1]
                c4 = max(ch[0] // 4, nk)
                for l, x in enumerate(ch):
                    _conv_spec(s, p + f".cv4.{l}.0", x, c4, 3)
                    _conv_spec(s, p + f".cv4.{l}.1", c4, c4, 3)
                    s[p + f".cv4.{l}.2.weight"] = ((nk, c4, 1, 1), "w")
                    s[p + f".cv4.{l}.2.bias"] = ((nk,), "b")
    return s


def bn_stats_path(scale, seed=7, pose=False):
    """Package data (lmx/data/): BatchNorm running statistics that belong to the synthetic weights of (scale, seed)."""
    import os

    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"yolov8{scale}{'-pose' if pose else ''}_bn_w{seed}.npz")


def synthetic_state_dict(cfg, seed, bn_stats=None):
    """Synthetic YOLOv8 weights: the seeded generator plus (optionally) BatchNorm running statistics (package data
    lmx/data/yolov8*_bn_w*.npz, produced once on the CPU by tests/golden/make_golden.py): random Conv+SiLU stacks
    are not stable over ~60 layers without the statistics a trained BatchNorm carries."""
    from . import weights

    sd = weights.synth_state_dict(param_spec(cfg), seed)
    if bn_stats is not None:
        stats = np.load(bn_stats) if isinstance(bn_stats, str) else bn_stats
        for k in stats.files if hasattr(stats, "files") else stats:
            if k.endswith("running_mean") or k.endswith("running_var"):
                if sd[k].shape != stats[k].shape:
                    raise ValueError(f"BN stat {k}: shape {stats[k].shape} != {sd[k].shape}")
                sd[k] = np.asarray(stats[k], np.float32)
    return sd


def fold_bn(sd, name):
    """ultralytics.utils.torch_utils.fuse_conv_and_bn: w' = w * g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps)."""
    w = sd[name + ".conv.weight"].astype(np.float32)
    g, b = sd[name + ".bn.weight"], sd[name + ".bn.bias"]
    mu, var = sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"]
    s = (g / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32)
    return (w * s[:, None, None, None]).astype(np.float32), (b - mu * s).astype(np.float32)


def count_params_flops(cfg, h=640, w=640):
    """Analytic parameter count (unfused, as Ultralytics reports) and MACs of the conv stack — cross-checks the
    restated architecture against the published 43.7 M / 165.2 GFLOPs (l) and 3.2 M / 8.7 GFLOPs (n)."""
    spec = param_spec(cfg)
    params = sum(int(np.prod(shape)) for name, (shape, _) in spec.items() if "running" not in name)
    params += REG_MAX  # Detect.dfl conv (frozen arange)
    macs = 0
    res = {}
    hw = (h, w)
    for i, m in enumerate(layer_table(cfg)):
        if m["kind"] == "conv":
            hw = ((hw[0] - 1) // 2 + 1, (hw[1] - 1) // 2 + 1)
            macs += hw[0] * hw[1] * m["c2"] * m["c1"] * 9
        elif m["kind"] == "c2f":
            c = m["c2"] // 2
            px = hw[0] * hw[1]
            macs += px * (m["c1"] * 2 * c + (2 + m["n"]) * c * m["c2"] + m["n"] * 2 * 9 * c * c)
        elif m["kind"] == "sppf":
            c_ = m["c1"] // 2
            macs += hw[0] * hw[1] * (m["c1"] * c_ + 4 * c_ * m["c2"])
        elif m["kind"] == "up":
            hw = (hw[0] * 2, hw[1] * 2)
        elif m["kind"] == "cat":
            hw = res[m["src"][1]]
        elif m["kind"] == "detect":
            c2 = max(16, m["ch"][0] // 4, REG_MAX * 4)
            c3 = max(m["ch"][0], min(cfg.nc, 100))
            for x, j in zip(m["ch"], m["src"]):
                px = res[j][0] * res[j][1]
                macs += px * (9 * x * c2 + 9 * c2 * c2 + c2 * 64 + 9 * x * c3 + 9 * c3 * c3 + c3 * cfg.nc)
                if cfg.kpt_shape is not None:
                    nk = cfg.kpt_shape[0] * cfg.kpt_shape[1]
                    c4 = max(m["ch"][0] // 4, nk)
                    macs += px * (9 * x * c4 + 9 * c4 * c4 + c4 * nk)
        res[i] = hw
    return params, macs


class _PlanF16:
    """Throughput plan: NHWC f16 activations, one launch per convolution (bias + SiLU + shortcut in the GEMM epilogue)."""
    cm = 1  # stored f16 channels per logical channel

    def __init__(self, det):
        self.w = det.w

    def stem(self, img):
        return K.stem_conv(img, *self.w["model.0"])

    def conv3(self, x, name, stride=1, res=None, out=None):
        return K.conv3x3(x, *self.w[name], act=K.ACT_SILU, stride=stride, res=res, out=out)

    def conv1(self, x, name, out=None, groups=None, g_out=None):
        return K.conv1x1(x, *self.w[name], act=K.ACT_SILU, out=out)

    def head(self, x, name, out=None):
        """the plain Conv2d(c, n_out, 1) that ends a Detect / Pose branch: f32 out, no activation"""
        return K.conv1x1(x, *self.w[name], act=K.ACT_NONE, out=out, out_dtype=torch.float32)

    pool5 = staticmethod(K.maxpool5)
    up2 = staticmethod(K.upsample2)


class _PlanExact:
    """Exact plan (csrc/exact.hip): activations travel as x3 triples, every convolution is ONE launch of the same GEMM /
    implicit-GEMM kernel over K' = 3K with f32 output, followed by lmx_k_split3 (activation, shortcut, re-split)."""
    cm = 3

    def __init__(self, det):
        self.det = det
        self.wx = {}

    def _w(self, name, groups=None):
        if name not in self.wx:
            dev = self.det.device
            w, b = self.det.wf32[name]
            if w.ndim == 4:  # [Cout, Cin, 3, 3] -> per tap [hi | mid | lo]
                co, ci = w.shape[:2]
                x3, sc, e = split_rows_x3(np.transpose(w, (0, 2, 3, 1)).reshape(co, 9 * ci), [ci] * 9)
            else:
                x3, sc, e = split_rows_x3(w, groups or [w.shape[1]])
            self.wx[name] = (torch.from_numpy(x3).to(dev), torch.from_numpy(np.ldexp(b, e).astype(np.float32)).to(dev),
                             torch.from_numpy(sc).to(dev))
        return self.wx[name]

    def stem(self, img):
        return K.stem_conv_x3(img, *self.det.w["model.0"])

    def conv3(self, x, name, stride=1, res=None, out=None):
        w, b, sc = self._w(name)
        n, H, W, cin = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        # a 3 x 3 convolution over K = 27 Cin is a handful of tiles per frame: split its k range over the idle CUs.  The factor
        # depends on the layer only (pixels of ONE frame), so a frame's bits do not depend on the batch it rides in
        sk = K.split_k_for(Ho * Wo, w.shape[0], w.shape[1], cin)
        t = K.conv3x3(x, w, b, act=K.ACT_NONE, stride=stride, scale=sc, out_dtype=torch.float32, split_k=sk)
        if out is None:
            out = torch.empty((n, Ho, Wo, 3 * w.shape[0]), dtype=torch.float16, device=x.device)
        return K.split3(t, K.ACT_SILU, out, res3=res)

    def conv1(self, x, name, out=None, groups=None, g_out=None):
        w, b, sc = self._w(name, groups)
        t = K.conv1x1(x, w, b, act=K.ACT_NONE, scale=sc, out_dtype=torch.float32)
        if out is None:
            out = torch.empty(t.shape[:3] + (3 * t.shape[3],), dtype=torch.float16, device=t.device)
        return K.split3(t, K.ACT_SILU, out, g=g_out)

    def head(self, x, name, out=None):
        w, b, sc = self._w(name)
        return K.conv1x1(x, w, b, act=K.ACT_NONE, scale=sc, out=out, out_dtype=torch.float32)

    pool5 = staticmethod(K.maxpool5_x3)
    up2 = staticmethod(K.upsample2)  # a copy: 3x the channels


PRECISIONS = ("f16", "exact")


class YoloDetector:
    """Device-resident fused YOLOv8.  ``detect(frames_bgr_u8)`` reproduces the predictor call per frame, batched.

    Two plans over the same kernels (``precision``):
      "exact" (default; services, adapters, the reference schedule): 22-bit operands through the f16 MFMA kernels (x3 format,
              csrc/exact.hip) — the prediction tensor agrees with the fp32 CPU path to ~1e-6, so NMS keep-sets and box
              indices are those of the fp32 path (north_star); 3x the MFMA work of
      "f16"   (the dense throughput schedule): f16 activations, scores within ~5e-3 of fp32.
    The plan is a property of the SCHEDULE, chosen by the caller — never of the batch size, so that a frame's result does
    not depend on how a clip is sharded over GPUs."""

    def __init__(self, cfg, state_dict, device="cuda", names=None, precision="exact"):
        if precision not in PRECISIONS:
            raise ValueError(f"precision {precision!r}: expected one of {PRECISIONS}")
        self.cfg = cfg
        self.precision = precision
        self.device = torch.device(device)
        self.names = {i: n for i, n in enumerate(names if names is not None else
                                                 (COCO_NAMES if cfg.nc == 80 else [f"class_{i}" for i in range(cfg.nc)]))}
        self.table = layer_table(cfg)
        self.nc_pad = (cfg.nc + 3) // 4 * 4
        sd = state_dict
        dev = self.device
        self.wf32 = {}  # name -> (folded f32 weight [Cout,Cin,3,3] or [Cout,Cin], f32 bias): the exact plan splits these lazily

        def pack3(name, wb=None):  # 3x3 conv -> [Cout, (ky,kx,ci)] f16 + f32 bias
            w, b = wb if wb is not None else fold_bn(sd, name)
            self.wf32[name] = (w, b)
            wp = np.transpose(w, (0, 2, 3, 1)).reshape(w.shape[0], -1)
            return (torch.from_numpy(np.ascontiguousarray(wp)).to(dev).half().contiguous(), torch.from_numpy(b).to(dev))

        def pack1(name, wb=None):
            if wb is None:
                w, b = fold_bn(sd, name)
                w = np.ascontiguousarray(w[:, :, 0, 0])
            else:
                w, b = wb
            self.wf32[name] = (w, b)
            return (torch.from_numpy(w).to(dev).half().contiguous(), torch.from_numpy(b).to(dev))

        self.w = {}
        for i, m in enumerate(self.table):
            p = f"model.{i}"
            if m["kind"] == "conv":
                if i == 0:
                    w, b = fold_bn(sd, p)  # stem stays f32: [ky][kx][c][Cout]
                    self.w[p] = (torch.from_numpy(np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))).to(dev),
                                 torch.from_numpy(b).to(dev))
                else:
                    self.w[p] = pack3(p)
            elif m["kind"] == "c2f":
                self.w[p + ".cv1"] = pack1(p + ".cv1")
                self.w[p + ".cv2"] = pack1(p + ".cv2")
                for j in range(m["n"]):
                    self.w[p + f".m.{j}.cv1"] = pack3(p + f".m.{j}.cv1")
                    self.w[p + f".m.{j}.cv2"] = pack3(p + f".m.{j}.cv2")
            elif m["kind"] == "sppf":
                self.w[p + ".cv1"] = pack1(p + ".cv1")
                self.w[p + ".cv2"] = pack1(p + ".cv2")
            elif m["kind"] == "detect":
                for l in range(3):
                    for br in ("cv2", "cv3"):
                        self.w[p + f".{br}.{l}.0"] = pack3(p + f".{br}.{l}.0")
                        self.w[p + f".{br}.{l}.1"] = pack3(p + f".{br}.{l}.1")
                        w = sd[p + f".{br}.{l}.2.weight"][:, :, 0, 0].astype(np.float32)
                        b = sd[p + f".{br}.{l}.2.bias"].astype(np.float32)
                        if br == "cv3" and self.nc_pad != cfg.nc:  # pad class rows to a multiple of 4 (GEMM N%4)
                            w = np.concatenate([w, np.zeros((self.nc_pad - cfg.nc, w.shape[1]), np.float32)], 0)
                            b = np.concatenate([b, np.zeros((self.nc_pad - cfg.nc,), np.float32)], 0)
                        self.w[p + f".{br}.{l}.2"] = pack1(p + f".{br}.{l}.2", (np.ascontiguousarray(w), b))
                if cfg.kpt_shape is not None:
                    # Pose branch: c4 = max(ch[0]//4, nk) is not a multiple of 8 in general (51 for 17x3 keypoints): zero-pad
                    # the channels (SiLU(0) = 0, so padded channels stay zero through the branch)
                    nk = cfg.kpt_shape[0] * cfg.kpt_shape[1]
                    c4 = max(m["ch"][0] // 4, nk)
                    c4p, self.nk_pad = (c4 + 7) // 8 * 8, (nk + 3) // 4 * 4

                    def padded3(name, cin_pad, cout_pad):
                        w, b = fold_bn(sd, name)
                        wz = np.zeros((cout_pad, cin_pad, 3, 3), np.float32)
                        wz[:w.shape[0], :w.shape[1]] = w
                        bz = np.zeros((cout_pad,), np.float32)
                        bz[:b.shape[0]] = b
                        return pack3(name, (wz, bz))

                    for l, x in enumerate(m["ch"]):
                        self.w[p + f".cv4.{l}.0"] = padded3(p + f".cv4.{l}.0", x, c4p)
                        self.w[p + f".cv4.{l}.1"] = padded3(p + f".cv4.{l}.1", c4p, c4p)
                        wz = np.zeros((self.nk_pad, c4p), np.float32)
                        wz[:nk, :c4] = sd[p + f".cv4.{l}.2.weight"][:, :, 0, 0]
                        bz = np.zeros((self.nk_pad,), np.float32)
                        bz[:nk] = sd[p + f".cv4.{l}.2.bias"]
                        self.w[p + f".cv4.{l}.2"] = pack1(p + f".cv4.{l}.2", (wz, bz))
        self._tabs = {}
        self._plans = {"f16": _PlanF16(self)}

    def _plan(self, precision):
        precision = precision or self.precision
        if precision not in PRECISIONS:
            raise ValueError(f"precision {precision!r}: expected one of {PRECISIONS}")
        if precision not in self._plans:
            self._plans[precision] = _PlanExact(self)
        return self._plans[precision]

    # ---- network --------------------------------------------------------------------------------------------
    def _c2f(self, P, i, m, x, out, groups=None):
        p = f"model.{i}"
        n, H, W, _ = x.shape
        c, cm = m["c2"] // 2, P.cm
        buf = torch.empty((n, H, W, cm * (2 + m["n"]) * c), dtype=torch.float16, device=x.device)
        P.conv1(x, p + ".cv1", out=buf[..., :cm * 2 * c], groups=groups, g_out=c)
        tmp = torch.empty((n, H, W, cm * c), dtype=torch.float16, device=x.device)
        for j in range(m["n"]):
            src = buf[..., cm * (1 + j) * c:cm * (2 + j) * c]
            dst = buf[..., cm * (2 + j) * c:cm * (3 + j) * c]
            P.conv3(src, p + f".m.{j}.cv1", out=tmp)
            P.conv3(tmp, p + f".m.{j}.cv2", res=src if m["shortcut"] else None, out=dst)
        return P.conv1(buf, p + ".cv2", out=out, groups=[c] * (2 + m["n"]))

    def _sppf(self, P, i, m, x, out):
        p = f"model.{i}"
        n, H, W, _ = x.shape
        c_, cm = m["c1"] // 2, P.cm
        buf = torch.empty((n, H, W, cm * 4 * c_), dtype=torch.float16, device=x.device)
        P.conv1(x, p + ".cv1", out=buf[..., :cm * c_])
        for j in range(3):
            P.pool5(buf[..., cm * j * c_:cm * (j + 1) * c_], buf[..., cm * (j + 1) * c_:cm * (j + 2) * c_])
        return P.conv1(buf, p + ".cv2", out=out, groups=[c_] * 4)

    def forward_letterboxed(self, img_u8, precision=None):
        """u8 RGB letterboxed [n,H,W,3] (H,W multiples of 32) -> pred f32 [n, A, 4+nc] (xywh in input pixels, scores)."""
        cfg, T = self.cfg, self.table
        P = self._plan(precision)
        cm = P.cm
        dev = img_u8.device
        n, H, W, _ = img_u8.shape
        f16 = torch.float16

        def buf(h, w, c):
            return torch.empty((n, h, w, cm * c), dtype=f16, device=dev)

        c_out = [m.get("c2", 0) for m in T]
        H8, W8, H16, W16, H32, W32 = H // 8, W // 8, H // 16, W // 16, H // 32, W // 32
        # concat buffers (producer slices): cat11 = [up(9) | 6], cat14 = [up(12) | 4], cat17 = [16 | 12], cat20 = [19 | 9]
        c4, c6, c9, c12 = c_out[4], c_out[6], c_out[9], c_out[12]
        c16, c19 = c_out[16], c_out[19]
        cat11 = buf(H16, W16, c9 + c6)
        cat14 = buf(H8, W8, c12 + c4)
        cat17 = buf(H16, W16, c16 + c12)
        cat20 = buf(H32, W32, c19 + c9)
        x = P.stem(img_u8)                                                                     # 0
        x = P.conv3(x, "model.1", stride=2)                                                    # 1
        x = self._c2f(P, 2, T[2], x, buf(H // 4, W // 4, c_out[2]))                            # 2
        x = P.conv3(x, "model.3", stride=2)                                                    # 3
        x4 = self._c2f(P, 4, T[4], x, cat14[..., cm * c12:])                                   # 4 -> cat14
        x = P.conv3(x4, "model.5", stride=2)                                                   # 5
        x6 = self._c2f(P, 6, T[6], x, cat11[..., cm * c9:])                                    # 6 -> cat11
        x = P.conv3(x6, "model.7", stride=2)                                                   # 7
        x = self._c2f(P, 8, T[8], x, buf(H32, W32, c_out[8]))                                  # 8
        x9 = self._sppf(P, 9, T[9], x, cat20[..., cm * c19:])                                  # 9 -> cat20
        P.up2(x9, cat11[..., :cm * c9])                                                        # 10, 11
        x12 = self._c2f(P, 12, T[12], cat11, cat17[..., cm * c16:], groups=[c9, c6])           # 12 -> cat17
        P.up2(x12, cat14[..., :cm * c12])                                                      # 13, 14
        p3 = self._c2f(P, 15, T[15], cat14, buf(H8, W8, c_out[15]), groups=[c12, c4])          # 15
        P.conv3(p3, "model.16", stride=2, out=cat17[..., :cm * c16])                           # 16, 17
        p4 = self._c2f(P, 18, T[18], cat17, buf(H16, W16, c_out[18]), groups=[c16, c12])       # 18
        P.conv3(p4, "model.19", stride=2, out=cat20[..., :cm * c19])                           # 19, 20
        p5 = self._c2f(P, 21, T[21], cat20, buf(H32, W32, c_out[21]), groups=[c19, c9])        # 21
        # Detect
        A = H8 * W8 + H16 * W16 + H32 * W32
        pred = torch.empty((n, A, 4 + cfg.nc), dtype=torch.float32, device=dev)
        a_off = 0
        kraw = []  # Pose: raw cv4 outputs per level, f32 [n,h,w,nk_pad]
        ldh = 64 + self.nc_pad
        for l, (feat, stride) in enumerate(((p3, 8), (p4, 16), (p5, 32))):
            p = f"model.22"
            h, w = feat.shape[1], feat.shape[2]
            head = torch.empty((n, h, w, ldh), dtype=torch.float32, device=dev)
            t = P.conv3(feat, p + f".cv2.{l}.0")
            t = P.conv3(t, p + f".cv2.{l}.1")
            P.head(t, p + f".cv2.{l}.2", out=head[..., :64])
            t = P.conv3(feat, p + f".cv3.{l}.0")
            t = P.conv3(t, p + f".cv3.{l}.1")
            P.head(t, p + f".cv3.{l}.2", out=head[..., 64:])
            K.detect_decode(head, pred, cfg.nc, stride, a_off)
            a_off += h * w
            if cfg.kpt_shape is not None:
                t = P.conv3(feat, p + f".cv4.{l}.0")
                t = P.conv3(t, p + f".cv4.{l}.1")
                kraw.append(P.head(t, p + f".cv4.{l}.2"))
        if cfg.kpt_shape is not None:
            return pred, kraw
        return pred

    # ---- pre / post ------------------------------------------------------------------------------------------
    def _letterbox_tables(self, sh, sw):
        key = (sh, sw)
        if key not in self._tabs:
            geo = LB.geometry(sh, sw, self.cfg.imgsz, 32, auto=True)
            tabs = None
            if (geo.rh, geo.rw) != (sh, sw):
                tabs = tuple(torch.from_numpy(t).to(self.device) for t in LB.resize_tables(sh, sw, geo.rh, geo.rw))
            self._tabs[key] = (geo, tabs)
        return self._tabs[key]

    def preprocess(self, frames_bgr):
        _, sh, sw, _ = frames_bgr.shape
        geo, tabs = self._letterbox_tables(sh, sw)
        return K.letterbox(frames_bgr, geo, tabs, swap_rb=True), geo

    def detect_pose(self, frames_bgr, conf=0.25, iou=0.7, max_det=300, precision=None):
        """Pose models: detect() plus keypoints f32 [n, max_det, K, ndim] in FRAME pixels (x, y, sigmoid visibility)."""
        if self.cfg.kpt_shape is None:
            raise ValueError("detect_pose: the model has no Pose head (YoloConfig.kpt_shape)")
        img, geo = self.preprocess(frames_bgr)
        pred, kraw = self.forward_letterboxed(img, precision)
        boxes, scores, cls, src, counts = K.nms(pred, conf, iou, max_det)
        K.scale_boxes(boxes, geo.pad_x, geo.pad_y, geo.gain, geo.sw, geo.sh)
        # ops.scale_coords subtracts the UNROUNDED padding (scale_boxes rounds it like LetterBox does)
        padx = (img.shape[2] - geo.sw * geo.gain) / 2
        pady = (img.shape[1] - geo.sh * geo.gain) / 2
        kpts = K.pose_gather(kraw, (8, 16, 32), src, counts, self.cfg.kpt_shape, padx, pady, geo.gain, geo.sw, geo.sh)
        return boxes, scores, cls, src, counts, kpts

    def detect(self, frames_bgr, conf=0.25, iou=0.7, max_det=300, precision=None):
        """u8 BGR [n,h,w,3] on device -> (boxes [n,max_det,4] xyxy in FRAME pixels, scores, cls, src, counts) on device.
        precision: None = the detector's default plan, or "exact" / "f16" (class docstring)."""
        img, geo = self.preprocess(frames_bgr)
        pred = self.forward_letterboxed(img, precision)
        if self.cfg.kpt_shape is not None:
            pred = pred[0]
        boxes, scores, cls, src, counts = K.nms(pred, conf, iou, max_det)
        K.scale_boxes(boxes, geo.pad_x, geo.pad_y, geo.gain, geo.sw, geo.sh)
        return boxes, scores, cls, src, counts
