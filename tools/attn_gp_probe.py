"""Long-sequence attention (Hiera's global blocks: B = 30 images, 8 heads, 4096 tokens, head dim 56): the software-pipelined kernel
(attn_gp_kernel, csrc/attn.hip) against the LDS-DMA form of attn_kernel it replaces (LMX_ATTN_NO_GP=1): same bits, time per launch.
Each variant runs in its own process (the switch is read once per process)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(30, 8, 4096, 56), (8, 12, 4096, 64), (2, 8, 4096, 56), (1, 2, 1153, 48), (2, 3, 257, 64), (3, 2, 256, 56), (1, 4, 320, 64), (1, 1, 4032, 56)]


def child():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
    import torch

    from lmx import kernels as K

    dev = torch.device("cuda:0")
    for B, H, T, hd in (SHAPES[:2] if os.environ.get("LMX_LIB") else SHAPES):  # decomposition builds: the two large shapes only
        D = H * hd
        g = torch.Generator().manual_seed(B * 1000 + T + hd)
        qkv = (torch.randn((B * T, 3 * D), generator=g) * 1.5).half().to(dev)
        out = torch.zeros((B * T, D), dtype=torch.float16, device=dev)
        run = lambda: K.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, B, H, T, T, hd, hd ** -0.5)  # noqa: E731
        run()
        torch.cuda.synchronize()
        digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
        if os.environ.get("LMX_GP_DUMP") and B * T <= 8192:
            import numpy as np
            np.save(f"{os.environ['LMX_GP_DUMP']}_{B}_{H}_{T}_{hd}.npy", out.cpu().numpy())
        n = 10 if B * H * T * T > 1e9 else 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        tf = 4.0 * B * H * T * T * 64 / us / 1e6  # MFMA work: head dim padded to 64
        print(f"B={B:3d} H={H:2d} T={T:4d} hd={hd:2d}  {us:9.1f} us  {tf:7.1f} TFLOP/s (64-wide)  sha {digest}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        res = {}
        for name, env in (("attn_gp_kernel (pipelined)", {}), ("attn_kernel LDS-DMA form (LMX_ATTN_NO_GP=1)", {"LMX_ATTN_NO_GP": "1"})):
            r = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env, "LMX_GP_DUMP": f"/tmp/gp{len(res)}"}, capture_output=True, text=True)
            print(f"# {name}\n{r.stdout}{r.stderr[-2000:] if r.returncode else ''}", flush=True)
            res[name] = [ln.split("sha ")[1] for ln in r.stdout.splitlines() if "sha " in ln]
        a, b = res.values()
        print("identical bits on every shape:", a == b and len(a) == len(SHAPES))
        import numpy as np
        for B, H, T, hd in SHAPES[2:]:
            x, y = (np.load(f"/tmp/gp{i}_{B}_{H}_{T}_{hd}.npy").astype(np.float64) for i in (0, 1))
            print(f"B={B} H={H} T={T} hd={hd}: max |difference| {np.abs(x - y).max():.3e}, differing elements {(x != y).mean():.4f}, max |value| {np.abs(x).max():.3f}")
