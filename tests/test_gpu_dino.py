"""End-to-end parity of the DINO path (services/dinov3-pipeline/app/main.py:95-115) on the GPU: raw BGR 1080p frames ->
HIP preprocessing -> HIP ViT -> mean-pooled embedding, against (a) the committed golden vectors produced by
transformers (tests/golden/make_golden.py) and (b) the fp32 oracle run here on the host, layer by layer on a small
config.  Bar (BASELINE.json north_star): embedding cosine >= 1 - 1e-4."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)


def test_small_dinov3_matches_oracle_per_token(cuda):
    from lmx import dino, synth, weights
    from oracle import preprocess as OP
    from oracle import vit

    cfg = dino.DinoConfig(hidden=256, layers=4, heads=4, mlp=1024, registers=4)
    sd = weights.synth_state_dict(dino.param_spec(cfg), seed=21)
    frames = np.stack([synth.synth_frame(9, i) for i in (0, 40)], 0)
    pv = torch.from_numpy(np.stack([OP.dino_pixel_values(f) for f in frames], 0))
    with torch.no_grad():
        ref = vit.dinov3_forward(cfg, sd, pv)
    m = dino.DinoEmbedder(cfg, sd, cuda)
    patches = m.preprocess(torch.from_numpy(frames).to(cuda))
    hs = m.hidden_states(patches, 2).cpu().view(2, cfg.tokens, cfg.hidden)
    err = (hs - ref).abs().max()
    assert float(err) < 3e-2, float(err)  # post-LayerNorm tokens are O(1); f16 operand rounding through 4 layers
    assert float(_cos(hs, ref).min()) > 1 - 1e-4
    emb = m.embed_frames(torch.from_numpy(frames).to(cuda)).cpu()
    assert float(_cos(emb, ref.mean(1)).min()) > 1 - 1e-4


def test_small_dinov2_matches_oracle(cuda):
    from lmx import dino, synth, weights
    from oracle import preprocess as OP
    from oracle import vit

    cfg = dino.DinoConfig(arch="dinov2", hidden=192, layers=3, heads=3, mlp=768, patch=14, registers=0, eps=1e-6,
                          pos_grid=37)
    sd = weights.synth_state_dict(dino.param_spec(cfg), seed=22)
    frames = np.stack([synth.synth_frame(10, i, 720, 1280) for i in (3, 77)], 0)
    pv = torch.from_numpy(np.stack([OP.dino_pixel_values(f) for f in frames], 0))
    with torch.no_grad():
        ref = vit.embed(cfg, sd, pv)
    emb = dino.DinoEmbedder(cfg, sd, cuda).embed_frames(torch.from_numpy(frames).to(cuda)).cpu()
    assert float(_cos(emb, ref).min()) > 1 - 1e-4
    assert float((emb - ref).abs().max()) < 2e-2


@pytest.mark.parametrize("name,mk", [("dinov3_vitl16_w3", "dinov3_vitl16"), ("dinov2_base_w4", "dinov2_base")])
def test_full_config_matches_golden(cuda, name, mk):
    """BASELINE cfg#4 architecture (DINOv3 ViT-L/16) and the reference-default one (DINOv2-B/14), from raw frames."""
    from lmx import dino, synth, weights

    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = getattr(dino, mk)()
    sd = weights.synth_state_dict(dino.param_spec(cfg), int(g["weight_seed"]))
    frames = np.stack([synth.synth_frame(int(g["clip_seed"]), int(i)) for i in g["frame_ids"]], 0)
    emb = dino.DinoEmbedder(cfg, sd, cuda).embed_frames(torch.from_numpy(frames).to(cuda)).cpu()
    ref = torch.from_numpy(g["embedding"])
    cos = _cos(emb, ref)
    print(name, "cos", cos.tolist(), "max abs", float((emb - ref).abs().max()))
    assert float(cos.min()) >= 1 - 1e-4, cos.tolist()


def test_cfg4_dinov3_vitl16_batch256(cuda):
    """BASELINE cfg#4: DINOv3 ViT-L/16 at batch 256 (one GPU takes the whole batch here; sharding over 8 GPUs is lmx.dist).
    Full path from 256 raw 1080p frames; the three golden frames sit at batch indices 0 / 128 / 255 and must match the
    committed transformers output (cosine >= 1 - 1e-4); embeddings do not depend on the batch they ride in (the same
    frames in a batch of 3 give identical bits) and a second run of the 256 batch is bit-identical."""
    from lmx import dino, synth, weights

    g = np.load(os.path.join(GOLD, "dinov3_vitl16_w3.npz"))
    cfg = dino.dinov3_vitl16()
    m = dino.DinoEmbedder(cfg, weights.synth_state_dict(dino.param_spec(cfg), int(g["weight_seed"])), cuda)
    gold_frames = [synth.synth_frame(int(g["clip_seed"]), int(i)) for i in g["frame_ids"]]
    pos = [0, 128, 255]
    base = np.stack([synth.synth_frame(21, i) for i in range(16)], 0)  # 16 distinct frames tiled over the other slots
    frames = torch.empty((256, 1080, 1920, 3), dtype=torch.uint8, device=cuda)
    d_base = torch.from_numpy(base).to(cuda)
    for i in range(0, 256, 16):
        frames[i:i + 16] = d_base
    for p_, f in zip(pos, gold_frames):
        frames[p_] = torch.from_numpy(f).to(cuda)
    emb = m.embed_frames(frames)
    torch.cuda.synchronize()
    assert tuple(emb.shape) == (256, 1024) and bool(torch.isfinite(emb).all())
    ref = torch.from_numpy(g["embedding"])
    cos = _cos(emb[pos].cpu(), ref)
    print("cfg4 b=256: golden frames at", pos, "cos", cos.tolist())
    assert float(cos.min()) >= 1 - 1e-4, cos.tolist()
    small = m.embed_frames(frames[pos].contiguous())
    assert torch.equal(small, emb[pos]), "embedding depends on the batch the frame rides in"
    assert torch.equal(emb[1], emb[17]) and torch.equal(emb[2], emb[242]), "identical frames in different slots differ"
    emb2 = m.embed_frames(frames)
    torch.cuda.synchronize()
    assert torch.equal(emb, emb2), "cfg#4 batch is not bit-reproducible"
