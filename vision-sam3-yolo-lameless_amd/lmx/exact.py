"""Host side of the exact plans (csrc/exact.hip): weights as two-term f16 splits for the x3 operand format.
Shared by lmx.yolo (YOLO's exact plan) and lmx.sam_decoder (the SAM mask decoder's)."""
import numpy as np


def split_rows_x3(w2, groups):
    """Exact plan, weight side (csrc/exact.hip): f32 rows w2 [N, K] whose K columns are `groups` consecutive channel groups ->
    (f16 [N, 3K] = per group [whi | whi / 2048 | wlo], scale f32 [N] = 2^-e, e) with whi + wlo the two-term f16 split of the row
    pre-scaled by 2^e (its largest weight lands in (2^13, 2^14]: wlo stays a normal f16 for every weight within 2^-12 of it)."""
    w2 = np.asarray(w2, np.float32)
    assert sum(groups) == w2.shape[1], (groups, w2.shape)
    amax = np.abs(w2).max(axis=1)
    e = np.where(amax > 0, 14 - np.ceil(np.log2(np.maximum(amax, 1e-30))), 0).astype(np.int32)
    ws = np.ldexp(w2, e[:, None]).astype(np.float32)
    whi = ws.astype(np.float16)
    wlo = (ws - whi.astype(np.float32)).astype(np.float16)
    wmid = (whi.astype(np.float32) / np.float32(2048)).astype(np.float16)
    parts, o = [], 0
    for g in groups:
        parts += [whi[:, o:o + g], wmid[:, o:o + g], wlo[:, o:o + g]]
        o += g
    return np.ascontiguousarray(np.concatenate(parts, 1)), np.ldexp(np.float32(1), -e).astype(np.float32), e
