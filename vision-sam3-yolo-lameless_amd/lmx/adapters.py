"""The Python seam of SURVEY.md §8(b): drop-in objects for the three third-party calls the services' glue code makes, so the
reference's own `main.py` files run on liblmx with their call sites unchanged (INTEGRATION.md shows the three-line patch).

  services/yolo-pipeline/app/main.py:24-37,76-96    YOLO(path) ; model(frame, verbose=False, conf=c) -> results;
                                                    result.boxes[i].xyxy[0] / .conf[0] / .cls[0] ; model.names[int]
  services/tleap-pipeline/app/main.py:142-171       ... and result.keypoints[j].data[0] for pose models
  services/sam3-pipeline/app/main.py:51-72,80-89    sam_model_registry[type](checkpoint=p) ; SamPredictor(sam) ;
                                                    predictor.set_image(frame) ; predictor.predict(point_coords=None,
                                                    point_labels=None, box=b[None, :], multimask_output=False)
                                                    -> (masks [1,H,W] bool, scores [1], low_res [1,256,256])
  services/dinov3-pipeline/app/main.py:34-36,98-113 AutoImageProcessor / AutoModel .from_pretrained(name) ;
                                                    inputs = processor(images=pil, return_tensors="pt") ;
                                                    {k: v.to(device)} ; model(**inputs).last_hidden_state [B,T,D]

Tensors handed back are torch tensors on the device (the glue calls .cpu().numpy() on them, as it does on Ultralytics');
numpy where segment_anything returns numpy.  Everything between the call and the return runs in liblmx kernels; there is
no CPU path (a missing library or a CPU-only torch raises).

Single-frame calls — the shape of every call the reference's loops make — can be replayed from a HIP graph captured on the first
call of an input shape (LMX_GRAPHS=1; lmx/graphs.py: same kernels, same bits, one launch instead of 200 - 700).  Measured, the
replay is no faster than the eager call (the latency is the GPU's chain of small kernels), so it is off by default; it frees the
host thread.  Calls with several frames always run eagerly."""
from pathlib import Path

import numpy as np
import torch

from . import checkpoints, dino, sam, sam_decoder, yolo
from .graphs import GraphedFn


def _device(device):
    if not torch.cuda.is_available():
        raise RuntimeError("lmx adapters need a GPU (liblmx has no CPU path)")
    return torch.device(device if device is not None else "cuda")


# ---- Ultralytics YOLO -------------------------------------------------------------------------------------------------------
class Boxes:
    """ultralytics.engine.results.Boxes, the slice the services use: len(), iteration / indexing into single-row Boxes,
    .xyxy [k,4], .conf [k], .cls [k] (torch tensors; row views keep the leading dimension, hence the glue's `[0]`)."""

    def __init__(self, xyxy, conf, cls):
        self.xyxy, self.conf, self.cls = xyxy, conf, cls

    def __len__(self):
        return int(self.xyxy.shape[0])

    def __getitem__(self, i):
        if isinstance(i, int):
            if not -len(self) <= i < len(self):
                raise IndexError(i)
            i = slice(i, i + 1) if i != -1 else slice(i, None)
        return Boxes(self.xyxy[i], self.conf[i], self.cls[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    @property
    def data(self):
        return torch.cat([self.xyxy, self.conf[:, None], self.cls[:, None].to(self.xyxy.dtype)], 1)


class Keypoints:
    """ultralytics Keypoints: indexing keeps the leading dimension; .data [k,K,3] (x, y, visibility), .xy, .conf."""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return int(self.data.shape[0])

    def __getitem__(self, i):
        if isinstance(i, int):
            i = slice(i, i + 1) if i != -1 else slice(i, None)
        return Keypoints(self.data[i])

    @property
    def xy(self):
        return self.data[..., :2]

    @property
    def conf(self):
        return self.data[..., 2] if self.data.shape[-1] > 2 else None


class Results:
    def __init__(self, boxes, names, orig_shape, keypoints=None):
        self.boxes, self.names, self.orig_shape, self.keypoints = boxes, names, orig_shape, keypoints

    def __len__(self):
        return len(self.boxes)


class LmxYolo:
    """`YOLO(weights)` replacement.  `weights`: a state-dict file (safetensors / weights_only .pt with Ultralytics' `model.N.*`
    names; lmx.checkpoints.load_yolo_weights), or a (YoloConfig, state dict) pair.  Calls take one HWC BGR uint8 frame (what
    cv2 hands the services), a list of frames, or a device uint8 tensor [n,h,w,3]; they return one Results per frame."""

    def __init__(self, weights, device=None, names=None):
        self.device = _device(device)
        if isinstance(weights, (str, Path)):
            cfg, sd = checkpoints.load_yolo_weights(weights)
        else:
            cfg, sd = weights
        self.detector = yolo.YoloDetector(cfg, sd, self.device, names=names)
        self.names = self.detector.names
        self.task = "pose" if cfg.kpt_shape is not None else "detect"
        self._graphs = {}  # (conf, iou, max_det) -> GraphedFn over one frame (the thresholds are kernel arguments of the capture)

    def to(self, device):  # the stock API allows model.to(...); weights already live on self.device
        return self

    def _frames(self, source):
        if isinstance(source, torch.Tensor):
            t = source if source.dim() == 4 else source[None]
            return t.to(self.device)
        if isinstance(source, (list, tuple)):
            return torch.from_numpy(np.stack([np.ascontiguousarray(f) for f in source], 0)).to(self.device)
        a = np.asarray(source)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"expected an HWC uint8 BGR frame, got {a.dtype} {a.shape}")
        return torch.from_numpy(np.ascontiguousarray(a)[None]).to(self.device)  # copies: the caller may reuse its buffer

    def __call__(self, source, verbose=False, conf=0.25, iou=0.7, max_det=300, **_unused):
        frames = self._frames(source)
        kp = None
        run = self.detector.detect_pose if self.task == "pose" else self.detector.detect
        if frames.shape[0] == 1:
            key = (float(conf), float(iou), int(max_det))
            g = self._graphs.get(key)
            if g is None and len(self._graphs) < 4:  # a service calls with ONE threshold set; a sweep over many runs eagerly
                g = self._graphs[key] = GraphedFn(lambda f, k=key: run(f, conf=k[0], iou=k[1], max_det=k[2]))
            out = g(frames) if g is not None else run(frames, conf=conf, iou=iou, max_det=max_det)
        else:
            out = run(frames, conf=conf, iou=iou, max_det=max_det)
        if self.task == "pose":
            boxes, scores, cls, _, counts, kp = out
        else:
            boxes, scores, cls, _, counts = out
        counts = counts.cpu().tolist()  # one host sync per call (the stock predictor syncs per box: yolo main.py:82-84)
        shape = tuple(frames.shape[1:3])
        out = []
        for j, k in enumerate(counts):
            out.append(Results(Boxes(boxes[j, :k], scores[j, :k], cls[j, :k].to(torch.float32)), self.names, shape,
                               Keypoints(kp[j, :k]) if kp is not None else None))
        return out

    predict = __call__


# ---- segment_anything -------------------------------------------------------------------------------------------------------
class LmxSam:
    """What `sam_model_registry[type](checkpoint=path)` returns: image encoder + prompt encoder / mask decoder on the device."""

    def __init__(self, cfg, state_dict, device=None):
        self.device = _device(device)
        self.cfg = cfg
        self.image_encoder = sam.SamVitEncoder(cfg, state_dict, self.device)
        self.mask_decoder = sam_decoder.MaskDecoder(state_dict, self.device, image_size=cfg.image, grid=cfg.grid)
        self.mask_threshold = 0.0

    def to(self, device):
        return self

    def eval(self):
        return self


def _sam_builder(model_type):
    def build(checkpoint=None, device=None):
        if checkpoint is None:
            raise ValueError("lmx has no randomly initialised SAM: pass checkpoint=<segment_anything .pth>")
        cfg, sd = checkpoints.load_sam_checkpoint(checkpoint, model_type)
        return LmxSam(cfg, sd, device)

    return build


sam_model_registry = {"default": _sam_builder("vit_h"), "vit_h": _sam_builder("vit_h"), "vit_l": _sam_builder("vit_l"),
                      "vit_b": _sam_builder("vit_b")}


class LmxSamPredictor:
    """`SamPredictor(sam)`: set_image caches the image embedding, predict decodes a prompt against it (the only state kept
    across calls, as in the reference: sam3 main.py:80-88).  `model` may also be any encoder with the HieraEncoder surface
    paired with a MaskDecoder (`LmxSamPredictor.from_parts`) — BASELINE's Hiera-B+ configuration."""

    def __init__(self, sam_model):
        self.model = sam_model
        self.encoder, self.decoder = sam_model.image_encoder, sam_model.mask_decoder
        self.device = sam_model.device
        self.is_image_set = False
        # the embedding is a static buffer of the capture (valid until the next set_image: the predictor's own contract); the
        # decoder's outputs are copied to the host before predict returns
        self._g_encode = GraphedFn(lambda f: self.encoder.encode(f)["fpn"][2], clone_outputs=False)
        self._g_decode = {}  # (original_size, input_size) -> GraphedFn(features, box)

    @classmethod
    def from_parts(cls, encoder, decoder):
        m = type("LmxSamParts", (), {})()
        m.image_encoder, m.mask_decoder, m.device, m.cfg = encoder, decoder, encoder.device, encoder.cfg
        return cls(m)

    def reset_image(self):
        self.is_image_set = False
        self.features = None
        self.original_size = self.input_size = None

    def set_image(self, image, image_format="RGB"):
        """image: HWC uint8.  Like segment_anything, the array is taken to be in `image_format` order and the model's
        normalisation constants are applied in that order — the service passes cv2's BGR frames as they are (Appendix C-2)."""
        a = np.asarray(image)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"set_image expects an HWC uint8 image, got {a.dtype} {a.shape}")
        if image_format not in ("RGB", "BGR"):
            raise ValueError(f"image_format must be 'RGB' or 'BGR', is {image_format}")
        if image_format == "BGR":  # segment_anything flips to the model's RGB order
            a = a[..., ::-1]
        frames = torch.from_numpy(np.ascontiguousarray(a)[None]).to(self.device)
        self.original_size = tuple(a.shape[:2])
        self.input_size = sam.resize_longest_side(a.shape[0], a.shape[1], self.encoder.cfg.image)
        e2 = self._g_encode(frames)
        self.features = e2.reshape(-1, e2.shape[-1])
        self.is_image_set = True

    def predict(self, point_coords=None, point_labels=None, box=None, mask_input=None, multimask_output=True, return_logits=False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        if point_coords is not None or point_labels is not None or mask_input is not None:
            raise NotImplementedError("lmx implements the box prompt the service uses (sam3 main.py:83-88)")
        if box is None:
            raise ValueError("predict needs box=np.ndarray [4] or [1,4] (xyxy, frame pixels)")
        if multimask_output:
            raise NotImplementedError("lmx decodes the single-mask output (multimask_output=False, sam3 main.py:87)")
        b = torch.from_numpy(np.asarray(box, np.float64).reshape(1, 4).astype(np.float32)).to(self.device)
        key = (self.original_size, self.input_size)
        g = self._g_decode.get(key)
        if g is None and len(self._g_decode) < 4:
            g = self._g_decode[key] = GraphedFn(lambda f, bx, k=key: self.decoder.predict(f, bx, k[0], k[1]), clone_outputs=False)
        out = g(self.features, b) if g is not None else self.decoder.predict(self.features, b, *key)
        low = out["lowres"].cpu().numpy()
        scores = out["iou"].cpu().numpy()
        if return_logits:
            raise NotImplementedError("return_logits=True: full-resolution logits are not materialised (the service thresholds)")
        masks = out["mask"].cpu().numpy().astype(bool)
        return masks, scores, low


SamPredictor = LmxSamPredictor


# ---- transformers AutoImageProcessor / AutoModel ----------------------------------------------------------------------------
class _Output:
    def __init__(self, last_hidden_state):
        self.last_hidden_state = last_hidden_state
        self.pooler_output = last_hidden_state[:, 0]


class LmxPixelValues(torch.Tensor):
    """The processor's "pixel_values": the normalised image already cut into the patch matrix the patch-embedding GEMM reads
    (f16 [B * patches, 3*P*P padded]) — the glue only forwards it (`model(**inputs)`), `.to(device)` included."""

    @staticmethod
    def wrap(t, batch):
        r = t.as_subclass(LmxPixelValues)
        r.lmx_batch = batch
        return r

    def to(self, *args, **kwargs):
        r = super().to(*args, **kwargs).as_subclass(LmxPixelValues)
        r.lmx_batch = self.lmx_batch
        return r


class LmxBatchFeature(dict):
    """transformers.BatchFeature, the slice the glue uses: a mapping (`model(**inputs)`) with `.to(device)` (main.py:107)."""

    def to(self, *args, **kwargs):
        return LmxBatchFeature({k: (v.to(*args, **kwargs) if hasattr(v, "to") else v) for k, v in self.items()})


class LmxDinoModel:
    """`AutoModel.from_pretrained(dir)` replacement: model(pixel_values=...) -> object with .last_hidden_state [B,T,D] f32."""

    def __init__(self, cfg, state_dict, device=None):
        self.device = _device(device)
        self.embedder = dino.DinoEmbedder(cfg, state_dict, self.device)
        self.config = cfg
        self._g_hidden = GraphedFn(lambda pv: self.embedder.hidden_states(pv, 1))
        self._g_pre = GraphedFn(lambda a: self.embedder.preprocess(a, rgb=True))

    @classmethod
    def from_pretrained(cls, model_dir, device=None):
        cfg, sd = checkpoints.load_dino_dir(model_dir)
        return cls(cfg, sd, device)

    def to(self, device):
        return self

    def eval(self):
        return self

    def __call__(self, pixel_values=None, **_unused):
        if not isinstance(pixel_values, LmxPixelValues):
            raise TypeError("pixel_values must come from LmxImageProcessor (the patch matrix liblmx reads)")
        B = pixel_values.lmx_batch
        pv = pixel_values.as_subclass(torch.Tensor)
        y = self._g_hidden(pv) if B == 1 else self.embedder.hidden_states(pv, B)
        return _Output(y.view(B, self.config.tokens, self.config.hidden))


class LmxImageProcessor:
    """`AutoImageProcessor.from_pretrained(dir)` replacement (dinov2-base preprocessing: bicubic shortest-edge 256, centre
    crop 224, /255, ImageNet mean/std — bit-exact against Pillow + BitImageProcessorPil, tests/test_oracle_preprocess.py).
    images: a PIL image or an HWC RGB uint8 array (or a list of them, equal sizes)."""

    def __init__(self, model):
        self.model = model

    @classmethod
    def from_pretrained(cls, model):
        return cls(model)

    def __call__(self, images=None, return_tensors="pt", **_unused):
        if return_tensors != "pt":
            raise ValueError("LmxImageProcessor returns torch tensors (return_tensors='pt')")
        items = images if isinstance(images, (list, tuple)) else [images]
        arr = np.stack([np.ascontiguousarray(np.asarray(im)) for im in items], 0)
        if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[3] != 3:
            raise ValueError(f"expected RGB uint8 image(s), got {arr.dtype} {arr.shape}")
        emb = self.model.embedder
        dev_arr = torch.from_numpy(arr).to(emb.device)
        patches = self.model._g_pre(dev_arr) if arr.shape[0] == 1 else emb.preprocess(dev_arr, rgb=True)
        return LmxBatchFeature({"pixel_values": LmxPixelValues.wrap(patches, arr.shape[0])})
