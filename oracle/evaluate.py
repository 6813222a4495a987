"""TEST INFRASTRUCTURE — numpy restatement of the reference's proposal-recall evaluator
(eval_recall/recall.py: bbox_overlaps 6-67, _recalls 70-100, eval_recalls 118-178).  Pinned:
tests/golden/make_golden.py imports the reference module itself (with a stand-in for the
``terminaltables`` pretty-printer it imports) and records its outputs; tests/test_cpu.py checks this
restatement against them bit for bit.  Only tests/, smoke() and bench.py's cpu_baseline may import it."""
from __future__ import annotations

import numpy as np


def bbox_overlaps(b1: np.ndarray, b2: np.ndarray, eps: float = 1e-6, legacy: bool = False) -> np.ndarray:
    ex = np.float32(1.0 if legacy else 0.0)
    b1, b2 = b1.astype(np.float32), b2.astype(np.float32)
    if b1.shape[0] * b2.shape[0] == 0:
        return np.zeros((b1.shape[0], b2.shape[0]), np.float32)
    a1 = (b1[:, 2] - b1[:, 0] + ex) * (b1[:, 3] - b1[:, 1] + ex)
    a2 = (b2[:, 2] - b2[:, 0] + ex) * (b2[:, 3] - b2[:, 1] + ex)
    xs = np.maximum(b1[:, None, 0], b2[None, :, 0]); ys = np.maximum(b1[:, None, 1], b2[None, :, 1])
    xe = np.minimum(b1[:, None, 2], b2[None, :, 2]); ye = np.minimum(b1[:, None, 3], b2[None, :, 3])
    ov = np.maximum(xe - xs + ex, np.float32(0)) * np.maximum(ye - ys + ex, np.float32(0))
    un = np.maximum(a1[:, None] + a2[None, :] - ov, np.float32(eps))
    return (ov / un).astype(np.float32)


def matched_ious(gts, proposals, proposal_nums, legacy=False) -> np.ndarray:
    total = sum(0 if g is None else g.shape[0] for g in gts)
    out = np.zeros((len(proposal_nums), total), np.float32)
    for k, pn in enumerate(proposal_nums):
        col = 0
        for g, p in zip(gts, proposals):
            ng = 0 if g is None else g.shape[0]
            if ng == 0:
                continue
            ious = bbox_overlaps(g, p[: min(p.shape[0], int(proposal_nums[-1])), :4], legacy=legacy)[:, :pn].copy()
            for j in range(ng):
                if ious.size == 0:
                    break
                rmax = ious.argmax(axis=1)
                vals = ious[np.arange(ng), rmax]
                gi = int(vals.argmax())
                out[k, col + j] = vals[gi]
                bi = rmax[gi]
                ious[gi, :] = -1
                ious[:, bi] = -1
            col += ng
    return out


def eval_recalls(gts, proposals, proposal_nums, iou_thrs, legacy=False) -> np.ndarray:
    proposal_nums, iou_thrs = np.asarray(proposal_nums), np.asarray(iou_thrs, dtype=np.float64)
    props = []
    for p in proposals:
        p = np.asarray(p)
        if p.ndim == 2 and p.shape[1] == 5:
            p = p[np.argsort(p[:, 4])[::-1], :]
        props.append(p)
    ious = matched_ious(gts, props, proposal_nums, legacy)
    total = ious.shape[1]
    ious = np.fliplr(np.sort(ious, axis=1))
    rec = np.zeros((proposal_nums.size, iou_thrs.size))
    for i, thr in enumerate(iou_thrs):
        rec[:, i] = (ious >= thr).sum(axis=1) / float(total)
    return rec
