"""GPU parity of the SAM image-encoder path (predictor.set_image, services/sam3-pipeline/app/main.py:80) through the
C-ABI: byte-exact preprocessing, per-kernel checks of the Hiera glue kernels, the Hiera trunk + FPN stage by stage
against the fp32 oracle (tiny config with every code path, and Hiera-B+ against the committed golden)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm()), float(torch.nn.functional.cosine_similarity(a, b, dim=0))


def test_glue_kernels(cuda):
    from lmx import kernels as K

    # 2x2 max pool, f16 slice and f32
    x = _rand((2, 8, 12, 48), 1).half()
    d = x.to(cuda)
    out = torch.zeros((2, 4, 6, 16), dtype=torch.float16, device=cuda)
    K.maxpool2(d[..., 16:32], out)
    ref = F.max_pool2d(x[..., 16:32].float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(out.float().cpu(), ref)
    xf = _rand((1, 6, 6, 8), 2)
    of = torch.zeros((1, 3, 3, 8), dtype=torch.float32, device=cuda)
    K.maxpool2(xf.to(cuda), of)
    assert torch.equal(of.cpu(), F.max_pool2d(xf.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1))
    # cast
    y = _rand((37, 112), 3, 50.0)
    assert torch.equal(K.cast_f16(y.to(cuda)).cpu(), y.half())
    # residual broadcast in the GEMM epilogue (position table)
    a, w = _rand((3 * 40, 64), 4).half(), _rand((32, 64), 5, 0.125).half()
    pos = _rand((40, 32), 6)
    got = K.gemm(a.to(cuda), w.to(cuda), res=pos.to(cuda), res_rows=40, out_dtype=torch.float32).cpu()
    ref = a.float() @ w.float().t() + pos.repeat(3, 1)
    assert float((got - ref).abs().max()) < 2e-4
    # im2col with normalisation LUT, canvas padding and conv padding
    img = np.random.default_rng(7).integers(0, 256, (2, 20, 28, 3), dtype=np.uint8)
    lut = _rand((3, 256), 8)
    cols = K.im2col_u8(torch.from_numpy(img).to(cuda), lut.to(cuda), 32, 32, 7, 7, 4, 3, 152).cpu()
    canvas = torch.zeros((2, 3, 32, 32))
    norm = torch.stack([lut[c][torch.from_numpy(img[..., c].astype(np.int64))] for c in range(3)], 1)
    canvas[:, :, :20, :28] = norm
    unf = F.unfold(canvas, 7, padding=3, stride=4)  # [2, 3*49, 64] with (c, ky, kx) order
    unf = unf.view(2, 3, 49, 64).permute(0, 3, 2, 1).reshape(2 * 64, 147)
    assert torch.equal(cols[:, :147], unf.half()) and float(cols[:, 147:].abs().max()) == 0


def test_sam_preprocess_bit_exact(cuda):
    from lmx import sam, synth, weights
    from oracle import preprocess as OP

    cfg = sam.HieraConfig(hidden=16, blocks=(1, 1, 1, 1), dims=(16, 32, 64, 128), heads=(1, 2, 4, 8), global_blocks=(),
                          pos_bkg=(7, 7), fpn_dim=32, image=1024)
    enc = sam.HieraEncoder(cfg, weights.synth_state_dict(sam.param_spec(cfg), 1), cuda)
    frames = np.stack([synth.synth_frame(13, 3), synth.synth_frame(13, 77)], 0)
    img, patches = enc.preprocess(torch.from_numpy(frames).to(cuda))
    for i in range(2):
        assert np.array_equal(img[i].cpu().numpy(), OP.sam_resized_u8(frames[i], 1024))
    pv = torch.from_numpy(OP.sam_pixel_values(frames[1], 1024))[None]
    unf = F.unfold(pv, 7, padding=3, stride=4).view(1, 3, 49, -1).permute(0, 3, 2, 1).reshape(-1, 147)
    assert torch.equal(patches[65536:, :147].cpu(), unf.half())


def _tiny(image=256):
    from lmx import sam

    return sam.HieraConfig(hidden=16, blocks=(1, 2, 3, 2), dims=(16, 32, 64, 128), heads=(1, 2, 4, 8), windows=(8, 4, 14, 7),
                           global_blocks=(4,), pos_bkg=(7, 7), fpn_dim=32, image=image)


@pytest.mark.parametrize("image", [256, 320])
def test_hiera_tiny_matches_oracle(cuda, image):
    """hd = 16 everywhere; padded windows (grids 16/8 with windows 14/7), Q-pool at three stage changes, a global block."""
    from lmx import sam, synth, weights
    from oracle import hiera as OH
    from oracle import preprocess as OP

    cfg = _tiny(image)
    sd = weights.synth_state_dict(sam.param_spec(cfg), seed=31)
    frames = np.stack([synth.synth_frame(14, i) for i in (5, 60)], 0)
    pv = torch.from_numpy(np.stack([OP.sam_pixel_values(f, image) for f in frames], 0))
    with torch.no_grad():
        ref_fpn, ref_stages = OH.encoder_forward(cfg, sd, pv)
    out = sam.HieraEncoder(cfg, sd, cuda).encode(torch.from_numpy(frames).to(cuda))
    torch.cuda.synchronize()
    for s, (got, ref) in enumerate(zip(out["stages"], ref_stages)):
        rel, cos = _rel(got.cpu(), ref)
        assert rel < 5e-3 and cos > 1 - 1e-4, f"stage {s}: rel {rel} cos {cos}"
    for l, (got, ref) in enumerate(zip(out["fpn"], ref_fpn)):
        rel, cos = _rel(got.float().cpu(), ref.permute(0, 2, 3, 1))
        assert rel < 5e-3 and cos > 1 - 1e-4, f"fpn {l}: rel {rel} cos {cos}"


def test_hiera_b_plus_matches_golden(cuda):
    """BASELINE cfg#3 architecture (Hiera-B+, 1024^2) from a raw 1080p frame vs the committed fp32 oracle output."""
    from lmx import sam, synth, weights

    g = np.load(os.path.join(GOLD, "hiera_bplus_w5.npz"))
    cfg = sam.hiera_b_plus()
    sd = weights.synth_state_dict(sam.param_spec(cfg), int(g["weight_seed"]))
    frames = np.stack([synth.synth_frame(int(g["clip_seed"]), int(i)) for i in g["frame_ids"]], 0)
    out = sam.HieraEncoder(cfg, sd, cuda).encode(torch.from_numpy(frames).to(cuda))
    torch.cuda.synchronize()
    emb = out["fpn"][2].float().cpu()[:, ::2, ::2]  # [n,64,64,256] image embedding level (fixture is subsampled)
    rel, cos = _rel(emb, torch.from_numpy(g["fpn2"]))
    last = out["stages"][3].cpu()[:, ::2, ::2]
    rel3, cos3 = _rel(last, torch.from_numpy(g["stage3"]))
    hi = out["fpn"][0].float().cpu()[:, ::16, ::16]
    relh, cosh = _rel(hi, torch.from_numpy(g["fpn0_sub"]))
    print("hiera-b+ vs fp32 oracle: fpn2 rel/cos", rel, cos, "stage3", rel3, cos3, "fpn0", relh, cosh)
    assert cos > 1 - 1e-4 and rel < 1e-2
    assert cos3 > 1 - 1e-4 and cosh > 1 - 1e-4


@pytest.mark.parametrize("precision", ["exact", "f16"])
def test_mask_decoder_matches_oracle(cuda, precision):
    """Prompt encoder + two-way decoder + upscaler + post-processing vs the fp32 oracle on the same embeddings/boxes.
    Bar (BASELINE.json north_star): mask IoU >= 0.999 — for both plans; the exact plan (f32 activations, 22-bit operands, f32
    attention: what the services run) must in addition reproduce the low-res logits to 1e-4 relative."""
    from lmx import kernels as K
    from lmx import sam_decoder
    from oracle import sam_decoder as OD

    sd = sam_decoder.synthetic_state_dict(41)
    rng = np.random.default_rng(3)
    n = 3
    # smooth embeddings (low-pass) so that the logits form blobs, not salt-and-pepper
    base = torch.from_numpy(rng.standard_normal((n, 256, 8, 8)).astype(np.float32))
    emb = F.interpolate(base, size=(64, 64), mode="bilinear", align_corners=False) * 2.0
    emb = (emb + 0.1 * torch.from_numpy(rng.standard_normal(emb.shape).astype(np.float32))).half().float()
    boxes = np.array([[300.0, 150.0, 1200.0, 900.0], [10.0, 20.0, 1900.0, 1000.0], [800.5, 400.25, 1000.0, 700.0]], np.float32)
    hw, rhw = (1080, 1920), (576, 1024)
    with torch.no_grad():
        sp = OD.prompt_encode_box(sd, torch.from_numpy(OD.scale_box(boxes, hw, rhw)))
        low_ref, iou_ref = OD.mask_decode(sd, emb, sp)
        mask_ref = OD.postprocess(low_ref, rhw, hw)
    dec = sam_decoder.MaskDecoder(sd, cuda, precision=precision)
    d_emb = emb.permute(0, 2, 3, 1).reshape(n * 4096, 256).contiguous().to(cuda)
    out = dec.predict(d_emb if precision == "exact" else d_emb.half(), torch.from_numpy(boxes).to(cuda), hw, rhw)
    torch.cuda.synchronize()
    sparse = K.prompt_box(torch.from_numpy(boxes).to(cuda), rhw[1] / hw[1], rhw[0] / hw[0], 1024.0, dec.gauss, dec.corner).cpu()
    assert float((sparse - sp).abs().max()) < 2e-4
    low = out["lowres"].cpu()
    rel = float((low - low_ref).norm() / low_ref.norm())
    print(precision, "decoder lowres rel err", rel, "iou head", out["iou"].cpu().tolist(), iou_ref.tolist())
    assert rel < (1e-4 if precision == "exact" else 2e-2)
    m = out["mask"].cpu().bool()
    for i in range(n):
        inter = float((m[i] & mask_ref[i]).sum())
        union = float((m[i] | mask_ref[i]).sum())
        iou = inter / union if union else 1.0
        frac = float(mask_ref[i].float().mean())
        print(f"mask {i}: IoU {iou:.6f}, coverage {frac:.3f}")
        assert 0.02 < frac < 0.98, "degenerate reference mask: the test would not measure anything"
        assert iou >= (0.9999 if precision == "exact" else 0.999), f"mask {i}: IoU {iou}"
    # mask_post alone on the oracle's logits: exact same pixels except where |value| ~ 0, and exact statistics
    mk, st = K.mask_post(low_ref.to(cuda), 1024, rhw[0], rhw[1], hw[0], hw[1])
    mk, st = mk.cpu().bool(), st.cpu()
    for i in range(n):
        diff = int((mk[i] ^ mask_ref[i]).sum())
        assert diff <= 20, f"mask_post differs from torch interpolate on {diff} pixels"
        ys, xs = torch.nonzero(mk[i], as_tuple=True)
        assert st[i, 0] == len(ys) and st[i, 1] == int(xs.sum()) and st[i, 2] == int(ys.sum())
        assert (st[i, 3], st[i, 4], st[i, 5], st[i, 6]) == (int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max()))


@pytest.mark.parametrize("hd", [64, 80])  # 80: SAM ViT-H, the attention kernel's wide head-dim class
def test_attention_relpos_bias(cuda, hd):
    """Decomposed relative-position bias (SAM v1): global 16x16 grid and 14x14 windows with zero-padded keys."""
    from lmx import kernels as K

    heads = 2
    D = heads * hd
    for (n, G, ws) in [(2, 16, 0), (1, 20, 14), (1, 64, 0)]:  # 64: the S == 64 fast path of the bias
        S = ws or G
        rows = n * G * G
        qkv = _rand((rows, 3 * D), 70, 1.0).half()
        bias = _rand((3 * D,), 71, 0.3).half()
        rh, rw = _rand((2 * S - 1, hd), 72, 0.3), _rand((2 * S - 1, hd), 73, 0.3)
        x = qkv.float().view(n, G, G, 3, heads, hd)
        if ws:
            pad = (-G) % ws
            full = bias.float().view(1, 1, 1, 3, heads, hd).expand(n, G + pad, G + pad, 3, heads, hd).clone()
            full[:, :G, :G] = x
            Gp = G + pad
            win = full.view(n, Gp // ws, ws, Gp // ws, ws, 3, heads, hd).permute(0, 1, 3, 2, 4, 5, 6, 7).reshape(-1, ws * ws, 3, heads, hd)
        else:
            win = x.reshape(n, G * G, 3, heads, hd)
        q, k, v = (win[:, :, i].permute(0, 2, 1, 3) for i in range(3))  # [B, heads, T, hd]
        idx = (torch.arange(S)[:, None] - torch.arange(S)[None, :]) + S - 1
        rq = q.reshape(-1, heads, S, S, hd)
        rel = (torch.einsum("bnhwc,hkc->bnhwk", rq, rh[idx])[..., :, None] + torch.einsum("bnhwc,wkc->bnhwk", rq, rw[idx])[..., None, :])
        a = (q * hd ** -0.5) @ k.transpose(-1, -2) + rel.reshape(q.shape[0], heads, S * S, S * S)
        o = (torch.softmax(a, -1) @ v).permute(0, 2, 1, 3).reshape(-1, S * S, D)
        if ws:
            o = o.view(n, Gp // ws, Gp // ws, ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(n, Gp, Gp, D)[:, :G, :G]
        ref = o.reshape(rows, D)
        d = qkv.to(cuda)
        bd = bias.to(cuda)
        out = torch.zeros((rows, D), dtype=torch.float16, device=cuda)
        if ws:
            nW = (Gp // ws) ** 2
            K.attention(d[:, :D], d[:, D:2 * D], d[:, 2 * D:], out, n * nW, heads, ws * ws, ws * ws, hd, hd ** -0.5,
                        window=dict(Gh=G, Gw=G, ws=ws, q_stride=1), pad_k=bd[D:2 * D], pad_v=bd[2 * D:],
                        rel_pos=(rh.to(cuda), rw.to(cuda)))
        else:
            K.attention(d[:, :D], d[:, D:2 * D], d[:, 2 * D:], out, n, heads, G * G, G * G, hd, hd ** -0.5,
                        rel_pos=(rh.to(cuda), rw.to(cuda)))
        err = float((out.float().cpu() - ref).abs().max())
        # f16 probabilities and f16 bias tables against an fp32 reference: the error grows with the number of keys
        assert err < (1e-2 if G * G > 1024 else 6e-3), f"rel-pos attention G={G} ws={ws}: max err {err}"


@pytest.mark.parametrize("hidden,heads", [(128, 2), (160, 2)])  # head dim 64 (vit_b / vit_l) and 80 (vit_h)
def test_sam_vit_encoder_matches_oracle(cuda, hidden, heads):
    """SAM v1 ImageEncoderViT (reference code path for sam_vit_* checkpoints): small config, windowed + global layers."""
    from lmx import sam, synth, weights
    from oracle import preprocess as OP
    from oracle import sam_vit as OV

    cfg = sam.SamVitConfig(hidden=hidden, layers=3, heads=heads, mlp=256, global_idx=(1,), window=14, image=512)
    sd = weights.synth_state_dict(sam.vit_param_spec(cfg), seed=61)
    frames = np.stack([synth.synth_frame(15, i) for i in (2, 33)], 0)
    pv = torch.from_numpy(np.stack([OP.sam_pixel_values(f, cfg.image) for f in frames], 0))
    with torch.no_grad():
        ref = OV.encoder_forward(cfg, sd, pv).permute(0, 2, 3, 1)
    out = sam.SamVitEncoder(cfg, sd, cuda).encode(torch.from_numpy(frames).to(cuda))["fpn"][2]
    torch.cuda.synchronize()
    rel, cos = _rel(out.float().cpu(), ref)
    print("sam vit small: rel", rel, "cos", cos)
    assert rel < 1e-2 and cos > 1 - 1e-4


def _mask_iou(a, b):
    inter = float((a & b).sum())
    union = float((a | b).sum())
    return inter / union if union else 1.0


@pytest.mark.parametrize("kind,fixture", [("hiera_bplus", "sam_mask_hiera_bplus_w5"), ("vit_b", "sam_mask_vit_b_w9")])
def test_mask_from_raw_frame_iou(cuda, kind, fixture):
    """north_star bar END TO END (services/sam3-pipeline/app/main.py:80-88: set_image + predict(box) -> masks[0]): raw
    1080p BGR frame + box -> HIP preprocessing -> HIP image encoder (Hiera-B+ = BASELINE cfg#3, SAM v1 ViT-B = the
    reference's own code path) -> HIP prompt encoder + mask decoder + post-processing, against the committed mask of the
    fp32 oracle run on the same frames (tests/golden/make_golden.py sam_masks).  Bar: mask IoU >= 0.999 for BOTH encoders on
    the plans the services run (default = "exact": the decoder with f32 activations and 22-bit operands; the ViT encoder with
    two-term weights).  Round 2 measured 0.99909 / 0.99821 for ViT-B on the all-f16 path; tools/sam_precision_probe.py
    (profiles/r03_sam_vit_precision_probe.txt) located the error in the decoder (8e-4 of the 9e-4 relative logit error) and
    the encoder's f16 WEIGHTS (3.9e-4), not in its activations."""
    from lmx import sam, sam_decoder, synth, weights

    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    seed = int(g["weight_seed"])
    frames = np.stack([synth.synth_frame(int(g["clip_seed"]), int(i)) for i in g["frame_ids"]], 0)
    h, w = frames.shape[1:3]
    if kind == "hiera_bplus":
        cfg = sam.hiera_b_plus()
        enc = sam.HieraEncoder(cfg, weights.synth_state_dict(sam.param_spec(cfg), seed), cuda)
    else:
        cfg = sam.sam_vit_b()
        enc = sam.SamVitEncoder(cfg, weights.synth_state_dict(sam.vit_param_spec(cfg), seed), cuda)
    dec = sam_decoder.MaskDecoder(sam_decoder.synthetic_state_dict(seed + 100), cuda)
    d_fr = torch.from_numpy(frames).to(cuda)
    e2 = enc.encode(d_fr)["fpn"][2]
    rhw = sam.resize_longest_side(h, w, 1024)
    out = dec.predict(e2.reshape(-1, e2.shape[-1]), torch.from_numpy(g["boxes"]).to(cuda), (h, w), rhw)
    torch.cuda.synchronize()
    ref = np.unpackbits(g["mask_bits"], axis=-1)[:, :, :w].astype(bool)
    got = out["mask"].cpu().numpy().astype(bool)
    emb_rel, emb_cos = _rel(e2.float().cpu()[:, ::4, ::4], torch.from_numpy(g["emb_sub"]).float())
    low_rel, _ = _rel(out["lowres"].cpu(), torch.from_numpy(g["lowres"]).float())
    ious = [_mask_iou(got[i], ref[i]) for i in range(len(frames))]
    print(f"{kind}: raw-frame mask IoU {ious}, coverage {ref.reshape(len(ref), -1).mean(1).tolist()}, embedding rel {emb_rel:.2e} cos {emb_cos:.8f}, "
          f"low-res logits rel {low_rel:.2e}, iou head {out['iou'].cpu().tolist()} vs {g['iou'].tolist()}")
    assert emb_cos > 1 - 1e-4
    for i, v in enumerate(ious):
        assert 0.02 < float(g["coverage"][i]) < 0.98
        assert v >= 0.999, f"{kind} frame {i}: raw-frame mask IoU {v}"
    # bit-packed output (what the service persists / gathers) decodes to the same mask
    from lmx import kernels as K
    bits = K.pack_bits(out["mask"]).cpu().numpy()
    assert np.array_equal(np.unpackbits(bits, axis=-1)[:, :, :w].astype(bool), got)


def test_hiera_fused_attention_halves_match_unfused_launches(cuda, monkeypatch):
    """Hiera-B+ at 1024^2 with the attention halves of stages 1 - 2 and of the block that opens stage 3 as single kernels
    (csrc/hiera.hip: lmx_k_hiera_attn8 with and without the LayerNorm inside, lmx_k_hiera_attn_pool in both shapes, lmx_k_hiera_attn4)
    against the same encoder on the separate launches
    they replace: every FPN level and stage output agrees to rounding (same rounding points, f32 sums in another order), and the
    fused entry points are the ones that ran."""
    from lmx import kernels as K
    from lmx import sam, synth, weights

    cfg = sam.hiera_b_plus()
    enc = sam.HieraEncoder(cfg, weights.synth_state_dict(sam.param_spec(cfg), 5), cuda)
    frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(2)], 0)).to(cuda)
    K.start_launch_trace()
    fused = enc.encode(frames)
    torch.cuda.synchronize()
    _, shapes = K.stop_launch_trace(by_shape=True)
    ran = sorted(key.split()[0] + ("/ln" if "ln_inside=1" in key else "") for (cls, key) in shapes if cls == "fused attention half")
    assert ran == ["hiera_attn4", "hiera_attn8", "hiera_attn8/ln", "hiera_attn_pool", "hiera_attn_pool"], ran  # (the pool form: stages 2 and 3)
    assert sum(r["launches"] for (cls, _), r in shapes.items() if cls == "fused attention half") == 6  # blocks 0 - 5
    for v in ("LMX_HIERA_ATTN8", "LMX_HIERA_ATTN4", "LMX_HIERA_ATTN_POOL"):
        monkeypatch.setenv(v, "0")
    K.start_launch_trace()
    plain = enc.encode(frames)
    torch.cuda.synchronize()
    _, shapes = K.stop_launch_trace(by_shape=True)
    assert not any(cls == "fused attention half" for (cls, _) in shapes)
    for name in ("fpn", "stages"):
        for lvl, (a, b) in enumerate(zip(fused[name], plain[name])):
            rel, cos = _rel(a.float().cpu(), b.float().cpu())
            print(f"fused vs unfused {name}[{lvl}]: rel {rel:.2e} cos-1 {cos - 1:.1e}")
            assert rel < 2e-3 and cos > 1 - 1e-5, (name, lvl, rel, cos)


def test_hiera_encoder_result_does_not_depend_on_the_batch(cuda):
    """A frame's Hiera-B+ output is bit-identical alone, in a batch of 3 and as the last frame of a batch of 7 (DESIGN.md section 3,
    Reproducibility): no launch of the encoder — the fused kernels of csrc/hiera.hip included — may pick a code path with a different
    summation order from the batch size.  This is what keeps a sharded clip's JSON equal to the unsharded one."""
    from lmx import sam, synth, weights

    cfg = sam.hiera_b_plus()
    enc = sam.HieraEncoder(cfg, weights.synth_state_dict(sam.param_spec(cfg), 5), cuda)
    frames = torch.from_numpy(np.stack([synth.synth_frame(3, 40 + i) for i in range(7)], 0)).to(cuda)
    alone = enc.encode(frames[6:7])
    in3 = enc.encode(frames[4:7])
    in7 = enc.encode(frames)
    torch.cuda.synchronize()
    for name in ("fpn", "stages"):
        for lvl in range(len(alone[name])):
            a = alone[name][lvl][0]
            assert torch.equal(a, in3[name][lvl][2]) and torch.equal(a, in7[name][lvl][6]), f"{name}[{lvl}] depends on the batch"
