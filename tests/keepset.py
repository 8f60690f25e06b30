"""Keep-set comparison rule of SURVEY.md section 7 ("ambiguity margin"), used by the YOLO parity tests.

north_star asks for bit-exact box indices and NMS keep-sets against the reference's fp32 CPU path.  A network that
stores f16 activations cannot promise that for candidates whose fp32 score lies within the network's own error of the
confidence threshold, or whose fp32 IoU with a kept box lies within that error of the IoU threshold.  The rule:

  run the greedy NMS of ultralytics.utils.ops.non_max_suppression (oracle/nms.py) on the fp32 prediction with THREE
  states per candidate - FIRM keep, FIRM drop, AMBIGUOUS - where a candidate is ambiguous iff some quantity deciding its
  fate (its score vs conf, its class argmax, its IoU with an earlier possibly-kept box vs iou, its rank against an
  overlapping box of nearly equal score) lies within (eps_score, eps_iou) of the decision boundary; ambiguity propagates
  (a box that only an ambiguous box would suppress is itself ambiguous).

  The device keep-set K must satisfy  FIRM  subset-of  K  subset-of  FIRM | AMBIGUOUS,  with equal class ids on FIRM:
  after removing the listed ambiguous candidates the two keep-sets (anchor indices) are IDENTICAL; any other
  difference fails.  eps_score / eps_iou are the measured deviations of the device prediction from the fp32 one on the
  same frame (measure_eps), printed by the tests and capped there.

tests/test_keepset_rule.py checks the rule itself: for random predictions and random perturbations bounded by eps, the
oracle NMS of the perturbed prediction always passes, and a dropped / added firm detection always fails."""
import numpy as np

MAX_WH = 7680.0


def _xyxy(box_xywh):
    b = np.asarray(box_xywh, np.float64)
    dw, dh = b[:, 2] / 2, b[:, 3] / 2
    return np.stack([b[:, 0] - dw, b[:, 1] - dh, b[:, 0] + dw, b[:, 1] + dh], 1)


def _iou_matrix(a, b):
    ax1, ay1, ax2, ay2 = (a[:, i][:, None] for i in range(4))
    bx1, by1, bx2, by2 = (b[:, i][None, :] for i in range(4))
    w = np.clip(np.minimum(ax2, bx2) - np.maximum(ax1, bx1), 0, None)
    h = np.clip(np.minimum(ay2, by2) - np.maximum(ay1, by1), 0, None)
    inter = w * h
    ua = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(ua > 0, inter / ua, 0.0)


def compact_pred(pred):
    """fp32 prediction [A, 4+nc] -> dict(box [A,4] xywh, score [A] best class score, cls [A], score2 [A] runner-up)."""
    p = np.asarray(pred, np.float32)
    cs = p[:, 4:]
    cls = cs.argmax(1)
    best = cs[np.arange(len(cs)), cls]
    if cs.shape[1] > 1:
        tmp = cs.copy()
        tmp[np.arange(len(cs)), cls] = -np.inf
        second = tmp.max(1)
    else:
        second = np.full_like(best, -np.inf)
    return dict(box=p[:, :4].copy(), score=best, cls=cls.astype(np.int16), score2=second)


def measure_eps(ref, dev, conf, window=0.15, iou_floor=0.3):
    """Measured deviation of the device prediction from the fp32 one on the anchors that can matter at `conf`
    (fp32 best score within `window` below conf, or above it): -> (eps_score, eps_iou, n_anchors_measured)."""
    sel = np.nonzero(np.maximum(ref["score"], dev["score"]) > conf - window)[0]
    if sel.size == 0:
        return 0.0, 0.0, 0
    eps_s = float(np.abs(ref["score"][sel].astype(np.float64) - dev["score"][sel]).max())
    eps_s = max(eps_s, float(np.abs(ref["score2"][sel].astype(np.float64) - dev["score2"][sel])[np.isfinite(ref["score2"][sel])].max(initial=0.0)))
    sel = sel[:4000]
    ir = _iou_matrix(_xyxy(ref["box"][sel]), _xyxy(ref["box"][sel]))
    idv = _iou_matrix(_xyxy(dev["box"][sel]), _xyxy(dev["box"][sel]))
    m = (ir > iou_floor) | (idv > iou_floor)
    eps_i = float(np.abs(ir - idv)[m].max()) if m.any() else 0.0
    return eps_s, eps_i, int(sel.size)


def classify(ref, conf, iou_thr, eps_s, eps_i, max_det=300):
    """Three-state greedy NMS on the fp32 compact prediction.  -> (firm: set of anchors, ambiguous: set of anchors,
    cls_of: {anchor: class id})."""
    score = ref["score"].astype(np.float64)
    cand = np.nonzero(score > conf - eps_s)[0]
    if cand.size == 0:
        return set(), set(), {}
    order = cand[np.argsort(-score[cand], kind="stable")]
    s = score[order]
    cls = ref["cls"][order].astype(np.int64)
    cls_amb = (s - ref["score2"][order].astype(np.float64)) <= 2 * eps_s   # the argmax class itself may differ on the device
    cand_amb = s <= conf + eps_s                                            # may or may not pass the confidence filter
    box = _xyxy(ref["box"][order])
    iou = _iou_matrix(box, box)
    same = (cls[:, None] == cls[None, :]) | cls_amb[:, None] | cls_amb[None, :]
    KEEP, AMB, SUPP = 1, 2, 3
    state = np.zeros(len(order), np.int8)
    n = len(order)
    for i in range(n):
        earlier = np.arange(n) < i
        # boxes that may come before i on the device although they follow it here: scores within 2*eps of i's
        near_after = (~earlier) & (np.arange(n) != i) & (s[i] - s <= 2 * eps_s) & (eps_s > 0)
        # a firm suppressor is firmly kept, firmly ahead of i in score order, of firmly the same class, firmly overlapping
        ahead = earlier & ((s - s[i] > 2 * eps_s) | (eps_s == 0))
        firm_sup = ahead & (state == KEEP) & (cls == cls[i]) & ~cls_amb & (iou[i] > iou_thr + eps_i)
        if firm_sup.any() and not cls_amb[i]:
            state[i] = SUPP
            continue
        maybe_sup = ((earlier & ((state == KEEP) | (state == AMB))) | near_after) & same[i] & (iou[i] > iou_thr - eps_i)
        if maybe_sup.any() or cand_amb[i] or cls_amb[i]:
            state[i] = AMB
        else:
            state[i] = KEEP
    firm = set(order[state == KEEP].tolist())
    amb = set(order[state == AMB].tolist())
    # the max_det cut (non_max_suppression keeps the first max_det survivors in score order): a survivor is firmly kept when
    # fewer than max_det survivors (firm or ambiguous) can precede it on the device (score > its score - 2 eps), firmly cut
    # when max_det FIRM survivors surely precede it (score > its score + 2 eps), ambiguous otherwise
    surv = [(a, float(score[a])) for a in order.tolist() if a in firm or a in amb]
    if len(surv) > max_det:
        sc_any = np.asarray([v for _, v in surv])
        sc_firm = np.asarray([v for a, v in surv if a in firm])
        for pos, (a, v) in enumerate(surv):
            if eps_s == 0:
                may_precede, surely_precede = pos, sum(1 for b, _ in surv[:pos] if b in firm)
            else:
                may_precede = int((sc_any > v - 2 * eps_s).sum()) - 1
                surely_precede = int((sc_firm > v + 2 * eps_s).sum())
            if surely_precede >= max_det:
                firm.discard(a)
                amb.discard(a)
            elif may_precede >= max_det and a in firm:
                firm.discard(a)
                amb.add(a)
    return firm, amb, {int(a): int(c) for a, c in zip(order, cls)}


def check_keepset(ref, conf, iou_thr, eps_s, eps_i, dev_src, dev_cls, label="", max_det=300):
    """Raises AssertionError unless FIRM <= device keep-set <= FIRM | AMBIGUOUS with equal classes on FIRM.
    -> dict(n_firm, n_ambiguous, n_dev, n_dev_ambiguous) for the report."""
    firm, amb, cls_of = classify(ref, conf, iou_thr, eps_s, eps_i, max_det)
    dev = [int(a) for a in np.asarray(dev_src).tolist()]
    dset = set(dev)
    assert len(dset) == len(dev), f"{label}: duplicate anchors in the device keep-set"
    missing = sorted(firm - dset)
    extra = sorted(dset - firm - amb)
    assert not missing, (f"{label}: {len(missing)} firm fp32 detections missing from the device keep-set (anchors {missing[:8]}, "
                         f"eps_score {eps_s:.2e}, eps_iou {eps_i:.2e})")
    assert not extra, (f"{label}: {len(extra)} device detections that the fp32 path firmly rejects (anchors {extra[:8]}, "
                       f"eps_score {eps_s:.2e}, eps_iou {eps_i:.2e})")
    for a, c in zip(dev, np.asarray(dev_cls).tolist()):
        if a in firm:
            assert int(c) == cls_of[a], f"{label}: anchor {a} kept with class {c}, fp32 says {cls_of[a]}"
    return dict(n_firm=len(firm), n_ambiguous=len(amb), n_dev=len(dev), n_dev_ambiguous=len(dset & amb))
