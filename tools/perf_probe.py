"""Quick per-kernel timing probe (development aid, not the graded bench): times the liblmx kernels at the shapes of
the BASELINE configs with HIP events on torch's current stream."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import _lib  # noqa: E402
if os.environ.get("LMX_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LMX_DBG_LIB"])
from lmx import dino, weights  # noqa: E402
from lmx import kernels as K  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["gemm", "dino"]
    if "gemm" in which:
        tot = 0.0
        # (M, N, K, act, f32-out-with-residual): DINOv3-L b=32 shapes, Hiera-B+ b=16 shapes, two square references
        for (M, N, K_, act, f32) in [(6432, 3072, 1024, 0, 0), (6432, 1024, 1024, 0, 1), (6432, 4096, 1024, 2, 0),
                                     (6432, 1024, 4096, 0, 1), (65536, 1792, 448, 2, 0), (65536, 448, 1792, 0, 1),
                                     (65536, 1344, 448, 0, 0), (65536, 448, 448, 0, 1), (1048576, 448, 112, 2, 0),
                                     (1048576, 112, 448, 0, 1), (1048576, 336, 112, 0, 0), (262144, 896, 224, 2, 0),
                                     (262144, 224, 896, 0, 1), (16384, 3584, 896, 2, 0), (16384, 896, 3584, 0, 1),
                                     (8192, 8192, 8192, 0, 0), (4096, 4096, 4096, 0, 0)]:
            a = torch.randn((M, K_), device=dev).half()
            w = (torch.randn((N, K_), device=dev) * K_ ** -0.5).half()
            b = torch.randn((N,), device=dev)
            if f32:
                out = torch.randn((M, N), device=dev, dtype=torch.float32)
                ms = timeit(lambda: K.gemm(a, w, bias=b, act=act, res=out, out=out), iters=10)
            else:
                out = torch.empty((M, N), device=dev, dtype=torch.float16)
                ms = timeit(lambda: K.gemm(a, w, bias=b, act=act, out=out), iters=10)
            tot += ms if M != 8192 and M != 4096 else 0
            print(f"gemm {M}x{N}x{K_} act{act} {'f32+res' if f32 else 'f16'}: {ms:.3f} ms  {2 * M * N * K_ / ms / 1e9:.1f} TFLOP/s",
                  flush=True)
        print(f"model-shape total {tot:.3f} ms", flush=True)
    if "narrow" in which:
        tot = 0.0
        for (M, N, K_, act, f32) in [(1048576, 448, 112, 2, 0), (1048576, 112, 448, 0, 1), (1048576, 336, 112, 0, 0),
                                     (1048576, 112, 112, 0, 1), (1048576, 672, 112, 0, 0), (262144, 896, 224, 2, 0),
                                     (262144, 224, 896, 0, 1), (262144, 672, 224, 0, 0), (262144, 224, 224, 0, 1),
                                     (65536, 448, 448, 0, 1), (65536, 1792, 448, 2, 0), (65536, 448, 1792, 0, 1)]:
            a = torch.randn((M, K_), device=dev).half()
            w = (torch.randn((N, K_), device=dev) * K_ ** -0.5).half()
            b = torch.randn((N,), device=dev)
            if f32:
                out = torch.randn((M, N), device=dev, dtype=torch.float32)
                ms = timeit(lambda: K.gemm(a, w, bias=b, act=act, res=out, out=out), iters=10)
                nbytes = M * K_ * 2 + 2 * M * N * 4
            else:
                out = torch.empty((M, N), device=dev, dtype=torch.float16)
                ms = timeit(lambda: K.gemm(a, w, bias=b, act=act, out=out), iters=10)
                nbytes = M * K_ * 2 + M * N * 2
            tot += ms
            print(f"gemm {M}x{N}x{K_} act{act} {'f32+res' if f32 else 'f16'}: {ms:.3f} ms  {2 * M * N * K_ / ms / 1e9:.1f} TFLOP/s  "
                  f"{nbytes / ms / 1e9:.2f} TB/s", flush=True)
        print(f"narrow total {tot:.3f} ms", flush=True)
    if "attn" in which or "dino" in which:
        B, H, T, hd = 256, 16, 201, 64
        qkv = torch.randn((B * T, 3 * H * hd), device=dev).half()
        o = torch.empty((B * T, H * hd), device=dev, dtype=torch.float16)
        D = H * hd
        ms = timeit(lambda: K.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, T, T, hd, 0.125))
        print(f"attention B{B} H{H} T{T}: {ms:.3f} ms  {4 * B * H * T * T * hd / ms / 1e9:.1f} TFLOP/s", flush=True)
    if "dino" in which:
        cfg = dino.dinov3_vitl16()
        t = time.time()
        sd = weights.synth_state_dict(dino.param_spec(cfg), 3)
        m = dino.DinoEmbedder(cfg, sd, dev)
        print(f"weights built+uploaded in {time.time() - t:.1f}s", flush=True)
        for B in (32, 256):
            patches = torch.randn((B * 196, 768), device=dev).half()
            ms = timeit(lambda: m.embed_patches(patches, B), iters=5, warm=2)
            print(f"dinov3-L embed_patches b={B}: {ms:.2f} ms  {B / ms * 1e3:.0f} frames/s  "
                  f"{125.7e9 * B / ms / 1e9:.0f} TFLOP/s", flush=True)
        frames = torch.randint(0, 256, (64, 1080, 1920, 3), dtype=torch.uint8, device=dev)
        ms = timeit(lambda: m.preprocess(frames), iters=5, warm=2)
        print(f"dino preprocess 64 x 1080p: {ms:.2f} ms  {64 * 1080 * 1920 * 3 / ms / 1e6:.0f} GB/s(in)", flush=True)


if __name__ == "__main__":
    main()
