"""Development aid for the wrong-load defect: mask_post with PLAIN loads (run with LMX_DBG_MASK=1)
in a loop on one stream, a background kernel in a loop on three other streams, nothing else.  Counts the launches whose
mask differs from the idle reference.

  LMX_LIB=$PWD/vision-sam3-yolo-lameless_amd/lmx/liblmx_dbg.so LMX_DBG_MASK=1 python tools/coresidency_probe.py {none|gemm|ln|attn}
(the DBG variants exist in the development build only: make -C vision-sam3-yolo-lameless_amd/csrc dbg; liblmx_dbg_noslp.so is
the same without hipcc's SLP vectoriser and is clean — round 2's decisive A/B, tools/defect_round2.sh)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-sam3-yolo-lameless_amd")]
from lmx import kernels as K  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
logits = torch.randn(16, 256, 256, device=dev, generator=g)
ref, ref_stats = K.mask_post(logits, 1024, 576, 1024, 1080, 1920)
torch.cuda.synchronize()
a = torch.randn(16384, 448, device=dev, generator=g).half()
w = torch.randn(1792, 448, device=dev, generator=g).half() * 0.05
x = torch.randn(262144, 112, device=dev, generator=g)
gam, bet = torch.ones(112, device=dev), torch.zeros(112, device=dev)
qkv = torch.randn(16 * 4096, 3 * 448, device=dev, generator=g).half()
ao = torch.empty(16 * 4096, 448, device=dev, dtype=torch.float16)


def background():
    if what == "gemm":
        K.gemm(a, w)
    elif what == "ln":
        K.layernorm(x, gam, bet, 1e-6)
    elif what == "attn":
        K.attention(qkv[:, :448], qkv[:, 448:896], qkv[:, 896:], ao, 16, 7, 4096, 4096, 64, 0.125)
    elif what.startswith("attn_"):  # other instantiations of the attention kernel: T, heads, head dim
        T, H, hd = {"attn_small": (16, 7, 64), "attn_w64": (64, 7, 64), "attn_w64o": (64, 8, 56), "attn_128": (128, 7, 64),
                    "attn_wide": (4096, 5, 80), "attn_1k": (1024, 7, 64)}[what]
        for _ in range(4 if T <= 128 else 1):
            K.attention(qkv[:, :H * hd], qkv[:, 448:448 + H * hd], qkv[:, 896:896 + H * hd], ao, 65536 // T, H, T, T, hd, 0.125)
    elif what.startswith("a128_"):  # the 128-query-tile kernel with one operand collapsed onto a single row (row stride 0)
        q_, k_, v_, o_ = qkv[:, :448], qkv[:, 448:896], qkv[:, 896:], ao
        if what == "a128_o0":
            o_ = ao[:1].expand(65536, 448)
        elif what == "a128_q0":
            q_ = qkv[:1, :448].expand(65536, 448)
        elif what == "a128_kv0":
            k_, v_ = qkv[:1, 448:896].expand(65536, 448), qkv[:1, 896:].expand(65536, 448)
        for _ in range(4):
            K.attention(q_, k_, v_, o_, 512, 7, 128, 128, 64, 0.125)
    elif what == "mm":  # a library GEMM: long-lived, LDS-heavy workgroups that are not ours
        torch.mm(big, big)


big = torch.randn(8192, 8192, device=dev, generator=g).half()
fgk = os.environ.get("FG", "mask")  # foreground kernel: mask (mask_post, fresh buffers), maskp (persistent buffers), ln
p_mask = torch.empty((16, 1080, 1920), dtype=torch.uint8, device=dev)
p_stats = torch.empty((16, 8), dtype=torch.int64, device=dev)
p_ws = torch.empty((16, 576, 1024), dtype=torch.float32, device=dev)
xf = torch.randn(1048576, 112, device=dev, generator=g)
ln_ref = K.layernorm(xf, gam, bet, 1e-6)
ln_out = torch.empty_like(ln_ref)
torch.cuda.synchronize()


def foreground():
    if fgk == "mask":
        m, _ = K.mask_post(logits, 1024, 576, 1024, 1080, 1920)
        return (m != ref).sum()
    if fgk == "maskp":
        K.check(K._lib.load().lmx_k_mask_post(K._ptr(logits), 16, 256, 1024, 576, 1024, 1080, 1920, K._ptr(p_mask), K._ptr(p_stats),
                                              K._ptr(p_ws), K._stream(logits.device)), "mask_post")
        return (p_mask != ref).sum()
    K.layernorm(xf, gam, bet, 1e-6, out=ln_out)
    return (ln_out != ln_ref).sum()


main = torch.cuda.current_stream()
fg = torch.cuda.Stream()
bgs = [torch.cuda.Stream() for _ in range(3)]
bad = torch.zeros((), dtype=torch.int64, device=dev)
launches_bad = torch.zeros((), dtype=torch.int64, device=dev)
for st in [fg] + bgs:
    st.wait_stream(main)
N = 150
for it in range(N):
    if what != "none":
        for st in bgs:
            with torch.cuda.stream(st):
                for _ in range(3 if what != "attn" else 1):
                    background()
    with torch.cuda.stream(fg):
        d = foreground()
        bad += d
        launches_bad += (d > 0).to(torch.int64)
torch.cuda.synchronize()
print(f"foreground={fgk} background={what} LMX_GEMM_V1={os.environ.get('LMX_GEMM_V1', '')} LMX_DBG_MASK={os.environ.get('LMX_DBG_MASK', '0')}: "
      f"{int(launches_bad)} of {N} foreground launches differ from the idle reference ({int(bad)} elements in all)", flush=True)
