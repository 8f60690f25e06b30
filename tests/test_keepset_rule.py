"""The keep-set margin rule (tests/keepset.py) checked against the NMS oracle on CPU: bounded perturbations of a
prediction always pass, unexplained differences always fail, and with eps = 0 the rule is exact equality."""
import numpy as np
import pytest

import keepset as KS
from oracle import nms as ONMS


def _random_pred(rng, n_clusters=12, per=14, nc=3):
    rows = []
    for _ in range(n_clusters):
        cx, cy = rng.uniform(80, 560, 2)
        w, h = rng.uniform(40, 200, 2)
        c = rng.integers(0, nc)
        for _ in range(per):
            sc = np.clip(rng.normal(0.5, 0.25), 0.01, 0.99)
            scores = rng.uniform(0.0, 0.05, nc)
            scores[c] = sc
            if rng.random() < 0.1:  # a close runner-up class now and then
                scores[(c + 1) % nc] = sc - rng.uniform(0, 0.02)
            rows.append([cx + rng.normal(0, 8), cy + rng.normal(0, 8), w * rng.uniform(0.85, 1.15), h * rng.uniform(0.85, 1.15), *scores])
    return np.asarray(rows, np.float32)


@pytest.mark.parametrize("max_det", [300, 12])
@pytest.mark.parametrize("seed", range(12))
def test_bounded_perturbations_pass(seed, max_det):
    rng = np.random.default_rng(seed)
    pred = _random_pred(rng)
    ref = KS.compact_pred(pred)
    for trial in range(6):
        d = pred.copy()
        d[:, 4:] += rng.uniform(-1, 1, d[:, 4:].shape).astype(np.float32) * np.float32(0.01 * (trial + 1) / 3)
        d[:, :4] += rng.uniform(-1, 1, d[:, :4].shape).astype(np.float32) * np.float32(0.4 * (trial + 1))
        dev = KS.compact_pred(d)
        for conf in (0.25, 0.5):
            eps_s, eps_i, _ = KS.measure_eps(ref, dev, conf, window=1.0, iou_floor=0.0)
            _, _, cls, src = ONMS.non_max_suppression(d, conf, max_det=max_det)
            rep = KS.check_keepset(ref, conf, 0.7, eps_s, eps_i, src, cls, f"seed {seed} trial {trial} conf {conf}", max_det=max_det)
            assert rep["n_firm"] + rep["n_ambiguous"] >= rep["n_dev"]


@pytest.mark.parametrize("seed", range(6))
def test_zero_eps_is_exact_equality(seed):
    rng = np.random.default_rng(100 + seed)
    pred = _random_pred(rng)
    ref = KS.compact_pred(pred)
    _, _, cls, src = ONMS.non_max_suppression(pred, 0.5)
    firm, amb, _ = KS.classify(ref, 0.5, 0.7, 0.0, 0.0)
    assert firm == set(src.tolist()) and not amb
    KS.check_keepset(ref, 0.5, 0.7, 0.0, 0.0, src, cls)
    # dropping, adding or re-labelling a detection is caught
    with pytest.raises(AssertionError):
        KS.check_keepset(ref, 0.5, 0.7, 0.0, 0.0, src[1:], cls[1:])
    other = next(a for a in range(len(pred)) if a not in set(src.tolist()))
    with pytest.raises(AssertionError):
        KS.check_keepset(ref, 0.5, 0.7, 0.0, 0.0, np.append(src, other), np.append(cls, 0))
    with pytest.raises(AssertionError):
        KS.check_keepset(ref, 0.5, 0.7, 0.0, 0.0, src, (cls + 1) % 3)


def test_small_eps_still_catches_a_firm_miss():
    rng = np.random.default_rng(7)
    pred = _random_pred(rng)
    ref = KS.compact_pred(pred)
    firm, amb, _ = KS.classify(ref, 0.5, 0.7, 5e-3, 1e-2)
    assert firm, "test needs firm detections"
    _, _, cls, src = ONMS.non_max_suppression(pred, 0.5)
    victim = next(iter(firm))
    keep = src != victim
    with pytest.raises(AssertionError):
        KS.check_keepset(ref, 0.5, 0.7, 5e-3, 1e-2, src[keep], cls[keep])


def test_max_det_cut():
    """More survivors than max_det: with eps = 0 the rule equals the truncated oracle NMS; with eps > 0 only survivors whose
    rank can straddle the cut become ambiguous."""
    rng = np.random.default_rng(5)
    n = 60
    pred = np.zeros((n, 4 + 2), np.float32)
    pred[:, 0] = 50 + 100 * np.arange(n)  # far apart: nothing suppresses anything
    pred[:, 1], pred[:, 2], pred[:, 3] = 50, 40, 40
    pred[:, 4] = np.linspace(0.9, 0.3, n).astype(np.float32)
    ref = KS.compact_pred(pred)
    firm, amb, _ = KS.classify(ref, 0.25, 0.7, 0.0, 0.0, max_det=20)
    _, _, cls, src = ONMS.non_max_suppression(pred, 0.25, max_det=20)
    assert firm == set(src.tolist()) and not amb
    firm, amb, _ = KS.classify(ref, 0.25, 0.7, 0.011, 0.0, max_det=20)  # score step is ~0.0102: ranks can move by two places
    assert set(range(0, 16)) <= firm and firm <= set(range(0, 20)) and amb and max(amb) <= 24 and min(amb) >= 16
