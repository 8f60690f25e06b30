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


SAM_PIXEL_MEAN = (123.675, 116.28, 103.53)
SAM_PIXEL_STD = (58.395, 57.12, 57.375)


def sam_resized_u8(frame, target=1024):
    """segment_anything ResizeLongestSide.apply_image: np.array(resize(to_pil_image(image), (nh, nw))) — torchvision's
    resize on a PIL image is PIL.Image.resize(..., BILINEAR).  The frame is used AS GIVEN (the sam3 service passes the
    cv2 BGR frame to set_image without conversion: services/sam3-pipeline/app/main.py:80,193,210)."""
    h, w = frame.shape[:2]
    scale = target * 1.0 / max(h, w)
    nh, nw = int(h * scale + 0.5), int(w * scale + 0.5)
    return np.asarray(Image.fromarray(np.ascontiguousarray(frame)).resize((nw, nh), Image.BILINEAR))


def sam_pixel_values(frame, target=1024):
    """SamPredictor.set_image -> Sam.preprocess: (x - pixel_mean) / pixel_std on the f32 image, zero pad to target^2.
    -> f32 [3, target, target]."""
    img = sam_resized_u8(frame, target).astype(np.float32)
    x = (img - np.array(SAM_PIXEL_MEAN, np.float32)) / np.array(SAM_PIXEL_STD, np.float32)
    out = np.zeros((target, target, 3), np.float32)
    out[:x.shape[0], :x.shape[1]] = x
    return np.ascontiguousarray(out.transpose(2, 0, 1))
