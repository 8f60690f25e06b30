"""oracle.preprocess + lmx.resample.norm_lut vs transformers' PIL image processor (the importable stand-in for
AutoImageProcessor(dinov2-base))."""
import numpy as np
import pytest
from PIL import Image

from lmx import resample as R
from oracle import preprocess as OP


def test_dino_pixel_values_match_bit_image_processor():
    tf = pytest.importorskip("transformers")
    try:
        from transformers.models.bit.image_processing_pil_bit import BitImageProcessorPil
    except Exception as e:  # pragma: no cover
        pytest.skip(f"BitImageProcessorPil not importable: {e}")
    proc = BitImageProcessorPil(size={"shortest_edge": 256}, crop_size={"height": 224, "width": 224},
                                image_mean=list(OP.IMAGENET_MEAN), image_std=list(OP.IMAGENET_STD), resample=3)
    frame = np.random.default_rng(5).integers(0, 256, (540, 960, 3), dtype=np.uint8)  # BGR
    ref = proc(images=Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1])), return_tensors="pt")["pixel_values"][0].numpy()
    got = OP.dino_pixel_values(frame)
    assert ref.shape == got.shape == (3, 224, 224)
    assert np.array_equal(ref, got), float(np.abs(ref - got).max())


def test_norm_lut_is_the_same_expression():
    frame = np.random.default_rng(6).integers(0, 256, (300, 400, 3), dtype=np.uint8)
    got = OP.dino_pixel_values(frame)
    img = OP.dino_resized_u8(frame)
    nh, nw = img.shape[:2]
    crop = img[(nh - 224) // 2:(nh - 224) // 2 + 224, (nw - 224) // 2:(nw - 224) // 2 + 224]
    lut = R.norm_lut(R.IMAGENET_MEAN, R.IMAGENET_STD)
    via_lut = np.stack([lut[c][crop[:, :, c]] for c in range(3)], 0)
    assert np.array_equal(via_lut, got)
