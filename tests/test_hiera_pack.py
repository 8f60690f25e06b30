"""Host-side packing of the fused Hiera attention kernels' operands (lmx/sam.py pack_hiera_attn / pack_hiera_attn4 /
pack_hiera_attn_pool; csrc/hiera.hip): the images the kernels read, addressed exactly as the kernels address them, give back the
torch-layout weights.  CPU only (the kernels themselves: tests/test_gpu_kernels.py::test_hiera_attn*)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vision-sam3-yolo-lameless_amd"))


def _kslot_feature(pos):
    """MFMA k-slot order (csrc/hiera.hip): position 32 s + 8 g + 4 h + i holds feature 16 (2 s + h) + 4 g + i."""
    s, r = divmod(pos, 32)
    g, r = divmod(r, 8)
    h, i = divmod(r, 4)
    return 16 * (2 * s + h) + 4 * g + i


def _frag256(img, row, ks, fg):
    """The 8 halfs a lane (row, fg) reads for k-step ks from an image with 256-byte rows: chunk (4 ks + fg) ^ (row & 15)."""
    c = ((4 * ks + fg) ^ (row & 15))
    return img[row * 128 + c * 8: row * 128 + c * 8 + 8]


def _frag128(img, row, s, fg):
    """... from an image with 128-byte rows: chunk (4 s + fg) ^ ((row >> 1) & 7)."""
    c = ((4 * s + fg) ^ ((row >> 1) & 7))
    return img[row * 64 + c * 8: row * 64 + c * 8 + 8]


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def test_kslot_order_is_a_permutation():
    assert sorted(_kslot_feature(p) for p in range(128)) == list(range(128))
    assert [_kslot_feature(p) for p in (0, 1, 4, 8, 32)] == [0, 1, 16, 4, 32]


def test_pack_hiera_attn_stage1():
    from lmx import sam

    D, heads, hd = 112, 2, 56
    wqkv, bqkv, wo, bo = _rand((3 * D, D), 1), _rand((3 * D,), 2), _rand((D, D), 3), _rand((D,), 4)
    for ln_inside in (False, True):
        wq, bq, wop, bop = sam.pack_hiera_attn(wqkv, bqkv, wo, bo, heads, ln_inside=ln_inside)
        assert wq.shape == (384, 128) and wq.dtype == np.float16 and wop.shape == (D, 128) and bq.shape == (384,)
        for sec in range(3):
            for hh in range(heads):
                r0 = sec * 128 + hh * 64
                src = wqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd].astype(np.float16)
                cols = [_kslot_feature(p) for p in range(128)] if ln_inside else list(range(128))
                for p, f in enumerate(cols):  # column p of the packed rows holds input feature f (zeros past D)
                    want = src[:, f] if f < D else np.zeros(hd, np.float16)
                    assert np.array_equal(wq[r0:r0 + hd, p], want)
                assert not wq[r0 + hd:r0 + 64].any()  # padding rows (v's row 63 included: the bias carries the 1)
                assert np.array_equal(bq[r0:r0 + hd], bqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd])
                assert bq[r0 + 63] == (1.0 if sec == 2 else 0.0) and not bq[r0 + hd:r0 + 63].any()
        for hh in range(heads):
            for p in range(64):
                d = _kslot_feature(p)
                want = wo[:, hh * hd + d].astype(np.float16) if d < hd else np.zeros(D, np.float16)
                assert np.array_equal(wop[:, 64 * hh + p], want)
        assert np.array_equal(bop, bo)


def test_pack_hiera_attn4_images():
    from lmx import sam

    D, heads, hd = 224, 4, 56
    wqkv, bqkv, wo, bo = _rand((3 * D, D), 5), _rand((3 * D,), 6), _rand((D, D), 7), _rand((D,), 8)
    img, bias = sam.pack_hiera_attn4(wqkv, bqkv, wo, bo, heads)
    assert img.shape == (16, 16384) and img.dtype == np.float16 and bias.shape == (heads * 192 + D,)
    for hh in range(heads):
        for sec in range(3):
            src = wqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd].astype(np.float16)
            # rows of 512 bytes: chunk c of row r at c ^ (r & 15); the kernel reads chunk 4 ks + fg of k-step ks < 7
            for row in (0, 1, 17, 55, 56, 63):
                for ks in range(7):
                    for fg in range(4):
                        c = (4 * ks + fg) ^ (row & 15)
                        got = img[4 * hh + sec][row * 256 + c * 8: row * 256 + c * 8 + 8]
                        want = src[row, 32 * ks + 8 * fg: 32 * ks + 8 * fg + 8] if row < hd else np.zeros(8, np.float16)
                        assert np.array_equal(got, want)
            assert np.array_equal(bias[hh * 192 + sec * 64: hh * 192 + sec * 64 + hd], bqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd])
        assert bias[hh * 192 + 128 + 63] == 1.0
        for row in (0, 3, 100, 223, 224, 255):
            for s in range(2):
                for fg in range(4):
                    got = _frag128(img[4 * hh + 3], row, s, fg)
                    want = np.array([wo[row, hh * hd + _kslot_feature(32 * s + 8 * fg + j)] if row < D and _kslot_feature(32 * s + 8 * fg + j) < hd else 0.0
                                     for j in range(8)], np.float32).astype(np.float16)
                    assert np.array_equal(got, want)
    assert np.array_equal(bias[heads * 192:], bo)


def test_pack_hiera_attn_pool_images():
    from lmx import sam

    Din, D, heads, hd = 112, 224, 4, 56
    wsc, bsc = _rand((D, Din), 9), _rand((D,), 10)
    wqkv, bqkv, wo, bo = _rand((3 * D, Din), 11), _rand((3 * D,), 12), _rand((D, D), 13), _rand((D,), 14)
    img, bias = sam.pack_hiera_attn_pool(wsc, bsc, wqkv, bqkv, wo, bo, heads)
    assert img.shape == (14, 16384) and bias.shape == (2 * D + heads * 192,)
    assert np.array_equal(bias[:D], bsc + bo)  # the shortcut's and the output projection's biases enter the same accumulators

    def check256(image, row, src_row):
        for ks in range(4):
            for fg in range(4):
                lo = 32 * ks + 8 * fg
                want = np.zeros(8, np.float16)
                if src_row is not None and lo < Din:
                    want = src_row[lo:lo + 8].astype(np.float16)
                assert np.array_equal(_frag256(image, row, ks, fg), want)

    for r in (0, 5, 127):
        check256(img[0], r, wsc[r])
    for r in (0, 95):
        check256(img[1], r, wsc[128 + r])
    check256(img[1], 96, None)  # rows 224.. of the shortcut image do not exist
    for hh in range(heads):
        for sec in range(2):
            for r in (0, 31, 55):
                check256(img[2 + 3 * hh], sec * 64 + r, wqkv[sec * D + hh * hd + r])
            check256(img[2 + 3 * hh], sec * 64 + 60, None)
        for r in (0, 55):
            check256(img[3 + 3 * hh], r, wqkv[2 * D + hh * hd + r])
        check256(img[3 + 3 * hh], 63, None)
        assert bias[D + hh * 192 + 128 + 63] == 1.0
        for sec in range(3):
            assert np.array_equal(bias[D + hh * 192 + sec * 64: D + hh * 192 + sec * 64 + hd], bqkv[sec * D + hh * hd: sec * D + (hh + 1) * hd])
        for row in (0, 111, 223, 230):
            for s in range(2):
                for fg in range(4):
                    want = np.array([wo[row, hh * hd + _kslot_feature(32 * s + 8 * fg + j)] if row < D and _kslot_feature(32 * s + 8 * fg + j) < hd else 0.0
                                     for j in range(8)], np.float32).astype(np.float16)
                    assert np.array_equal(_frag128(img[4 + 3 * hh], row, s, fg), want)
