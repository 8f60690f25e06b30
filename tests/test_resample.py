"""lmx.resample (host tables the device resize kernels consume) vs the installed Pillow, bit for bit."""
import numpy as np
import pytest
from PIL import Image

from lmx import resample as R


@pytest.mark.parametrize("shape,out,filt,pil", [
    ((1080, 1920), (455, 256), R.BICUBIC, Image.BICUBIC),     # DINO shortest-edge 256 on a 1080p frame
    ((720, 1280), (455, 256), R.BICUBIC, Image.BICUBIC),      # the reference's canonical clips are 1280x720
    ((1080, 1920), (1024, 576), R.BILINEAR, Image.BILINEAR),  # SAM ResizeLongestSide(1024)
    ((100, 130), (333, 256), R.BICUBIC, Image.BICUBIC),       # upscale
    ((64, 64), (64, 32), R.BILINEAR, Image.BILINEAR),         # one axis unchanged
])
def test_tables_reproduce_pillow(shape, out, filt, pil):
    img = np.random.default_rng(3).integers(0, 256, shape + (3,), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize(out, pil))
    got = R.resize_u8_reference(img, out[0], out[1], filt)
    assert np.array_equal(ref, got)


def test_shortest_edge_size():
    assert R.shortest_edge_size(1080, 1920, 256) == (256, 455)
    assert R.shortest_edge_size(1920, 1080, 256) == (455, 256)
    assert R.shortest_edge_size(720, 1280, 256) == (256, 455)
    assert R.shortest_edge_size(256, 300, 256) == (256, 300)
