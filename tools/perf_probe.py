"""Quick per-kernel timing probe (development aid, not the graded bench): times the liblmx kernels at the shapes of
the BASELINE configs with HIP events on torch's current stream."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
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
        for (M, N, K_) in [(51456, 3072, 1024), (51456, 1024, 1024), (51456, 4096, 1024), (51456, 1024, 4096),
                           (8192, 8192, 8192), (4096, 4096, 4096)]:
            a = torch.randn((M, K_), device=dev).half()
            w = torch.randn((N, K_), device=dev).half()
            out = torch.empty((M, N), device=dev, dtype=torch.float16)
            ms = timeit(lambda: K.gemm(a, w, out=out))
            print(f"gemm {M}x{N}x{K_}: {ms:.3f} ms  {2 * M * N * K_ / ms / 1e9:.1f} TFLOP/s", flush=True)
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
