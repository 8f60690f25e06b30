"""oracle.preprocess — the services' frame preprocessing, run through the real libraries on the CPU.
TEST INFRASTRUCTURE (see oracle/__init__.py).

dino_pixel_values follows services/dinov3-pipeline/app/main.py:95-107: cv2.cvtColor(BGR2RGB) -> PIL.Image.fromarray ->
AutoImageProcessor(dinov2-base preprocessor_config: shortest_edge 256 BICUBIC, center crop 224, rescale 1/255,
ImageNet mean/std).  The resize is Pillow itself; crop/rescale/normalize restate transformers.image_transforms
(center_crop :~460, rescale :89-124, normalize :384-442) and are checked against BitImageProcessorPil in
tests/test_oracle_preprocess.py."""
import numpy as np
from PIL import Image

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def shortest_edge_size(h, w, edge):
    short, long = (w, h) if w <= h else (h, w)
    if short == edge:
        return h, w
    new_short, new_long = edge, int(edge * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def dino_resized_u8(frame_bgr, edge=256):
    rgb = np.ascontiguousarray(frame_bgr[:, :, ::-1])  # cv2.COLOR_BGR2RGB is a channel reversal
    h, w = rgb.shape[:2]
    nh, nw = shortest_edge_size(h, w, edge)
    return np.asarray(Image.fromarray(rgb).resize((nw, nh), Image.BICUBIC))


def dino_pixel_values(frame_bgr, edge=256, crop=224):
    """-> f32 [3, crop, crop] (CHW, as the processor returns)."""
    img = dino_resized_u8(frame_bgr, edge)
    nh, nw = img.shape[:2]
    top, left = (nh - crop) // 2, (nw - crop) // 2
    img = img[top:top + crop, left:left + crop]
    x = (img.astype(np.float64) * (1 / 255)).astype(np.float32)
    x = (x - np.array(IMAGENET_MEAN, np.float32)) / np.array(IMAGENET_STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
