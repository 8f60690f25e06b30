"""Synthetic 1080p clips (BASELINE cfg#5: 5 s @ 30 fps = 150 frames of [1080,1920,3] u8): a textured
rectangle moving over a low-pass noise background, BGR like cv2 frames.  Deterministic in (seed, frame index), pure
numpy, so the container that writes the golden vectors and the GPU box that checks them see identical bytes."""
import numpy as np


def _background(rng, h, w):
    # low-pass noise: random 1/8-resolution field, nearest-upsampled, plus fine noise
    small = rng.integers(40, 200, (h // 8 + 1, w // 8 + 1, 3), dtype=np.int32)
    big = np.repeat(np.repeat(small, 8, 0), 8, 1)[:h, :w]
    fine = rng.integers(-12, 13, (h, w, 3), dtype=np.int32)
    return np.clip(big + fine, 0, 255).astype(np.uint8)


def synth_frame(seed, idx, h=1080, w=1920, background=None):
    """`background`: the clip's background if the caller already has it (it depends on the seed only; synth_clip reuses it)."""
    if background is None:
        rng = np.random.default_rng([int(seed), 7919])
        background = _background(rng, h, w)
    img = background.copy()
    # moving textured box ("cow"): ~40% of the height, walks left to right
    bh, bw = int(h * 0.42), int(w * 0.30)
    x0 = int((w - bw) * ((idx % 150) / 149.0))
    y0 = int(h * 0.35 + h * 0.03 * np.sin(idx / 7.0))
    yy, xx = np.mgrid[0:bh, 0:bw]
    tex = (96 + 64 * np.sin(xx / 9.0 + idx * 0.1) * np.cos(yy / 13.0)).astype(np.int32)
    patch = np.stack([tex + 30, tex, tex - 30], -1)
    spots = ((xx // 24 + yy // 24) % 3 == 0)[..., None] * 50
    img[y0:y0 + bh, x0:x0 + bw] = np.clip(patch + spots, 0, 255).astype(np.uint8)
    return img


def synth_clip(seed, n_frames=150, h=1080, w=1920, start=0):
    bg = _background(np.random.default_rng([int(seed), 7919]), h, w)
    out = np.empty((n_frames, h, w, 3), np.uint8)
    for i in range(n_frames):
        out[i] = synth_frame(seed, start + i, h, w, background=bg)
    return out


def cfg1_frame(size=640):
    """BASELINE cfg#1 input: `np.random.default_rng(0).integers(0, 256, (640, 640, 3), uint8)`."""
    return np.random.default_rng(0).integers(0, 256, (size, size, 3), dtype=np.uint8)


def cfg2_frames(n=32, seed=1, size=640):
    """BASELINE cfg#2 input: uniform u8 noise + 8 pasted solid rectangles per image, BGR u8 [n,640,640,3]."""
    rng = np.random.default_rng(seed)
    fr = rng.integers(0, 256, (n, size, size, 3), dtype=np.uint8)
    for i in range(n):
        for _ in range(8):
            x0, y0 = (int(v) for v in rng.integers(0, size - 64, 2))
            w, h = (int(v) for v in rng.integers(48, 320, 2))
            fr[i, y0:min(size, y0 + h), x0:min(size, x0 + w)] = rng.integers(0, 256, 3, dtype=np.uint8)
    return fr
