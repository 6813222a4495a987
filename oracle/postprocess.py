"""TEST INFRASTRUCTURE — CPU oracle for score filter / top-k / NMS / gather / retrieval.

numpy restatement (integer/index work, bit-exact contract).  The defined total order
is the one SURVEY.md §7 fixes: candidates enumerate row-major over [anchor, class]
(``torch.nonzero`` order, generate_proposal.py:110-111); sorting is by score
descending with ties broken by candidate index ascending (= the reference run with
``sort(stable=True)``; its default unstable sort is not reproducible at tie level).

Reference sites:
  filter_scores_and_topk           generate_proposal.py:85-131
  head_predict per-image loop      generate_proposal.py:1196-1218; extract_embedding.py:1233-1260
  predict_by_feat per-image loop   wedetect/models/dense_heads/yolo_world_head.py:680-748
  un-letterbox + clamp             generate_proposal.py:1106-1115
  retrieval similarity             eval_retrieval/retrieval_metric.py:367-377
  batched NMS: torchvision.ops.batched_nms (call generate_proposal.py:1210) and
  mmcv.ops.batched_nms (call via _bbox_post_process, yolo_world_head.py:740-744) —
  third-party native code, absent here: **parity unpinned**.  Restated from their
  documented behaviour: per class, greedy in descending score; box i suppresses a
  later box j of the same class iff  inter / (area_i + area_j - inter) > thr  with
  area = (x2-x1)*(y2-y1), inter = max(0, .)*max(0, .), fp32, evaluated left to right;
  kept indices are returned in descending-score order.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

f32 = np.float32


def filter_scores_and_topk(scores: np.ndarray, score_thr: float, topk: int):
    """scores [N, K] fp32 -> (scores[n], labels[n] i64, anchor_idx[n] i64), n <= topk."""
    scores = np.ascontiguousarray(scores, dtype=f32)
    n, k = scores.shape
    flat = scores.reshape(-1)
    valid = np.nonzero(flat > f32(score_thr))[0]          # ascending flat index
    s = flat[valid]
    order = np.argsort(-s, kind="stable")                  # desc, ties: index asc
    num = min(int(topk), valid.shape[0])
    order = order[:num]
    idx = valid[order]
    return s[order], (idx % k).astype(np.int64), (idx // k).astype(np.int64)


def iou_suppresses(bi: np.ndarray, bj: np.ndarray, thr: float) -> np.ndarray:
    """fp32 IoU test of one box ``bi`` [4] against boxes ``bj`` [m,4]; True = suppressed."""
    area_i = (bi[2] - bi[0]) * (bi[3] - bi[1])
    area_j = (bj[:, 2] - bj[:, 0]) * (bj[:, 3] - bj[:, 1])
    xx1 = np.maximum(bi[0], bj[:, 0])
    yy1 = np.maximum(bi[1], bj[:, 1])
    xx2 = np.minimum(bi[2], bj[:, 2])
    yy2 = np.minimum(bi[3], bj[:, 3])
    w = np.maximum(f32(0), xx2 - xx1)
    h = np.maximum(f32(0), yy2 - yy1)
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        ovr = inter / (area_i + area_j - inter)
    return ovr > f32(thr)


def _iou_f32(bi: np.ndarray, bj: np.ndarray) -> np.ndarray:
    area_i = (bi[2] - bi[0]) * (bi[3] - bi[1])
    area_j = (bj[:, 2] - bj[:, 0]) * (bj[:, 3] - bj[:, 1])
    w = np.maximum(f32(0), np.minimum(bi[2], bj[:, 2]) - np.maximum(bi[0], bj[:, 0]))
    h = np.maximum(f32(0), np.minimum(bi[3], bj[:, 3]) - np.maximum(bi[1], bj[:, 1]))
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area_i + area_j - inter)


def batched_nms(boxes: np.ndarray, scores: np.ndarray, labels: np.ndarray, thr: float,
                max_keep: Optional[int] = None, stats: Optional[dict] = None) -> np.ndarray:
    """Class-aware greedy NMS.  Inputs must already be in the defined total order
    (they are: the output of filter_scores_and_topk).  Returns kept candidate indices
    in that order.  ``max_keep`` stops early — identical to slicing the full result,
    because whether a box is kept depends only on earlier kept boxes.

    ``stats`` (a dict, filled in place) records how far every decision taken was from flipping — what a second
    implementation with ~1e-6 numerical noise on scores / boxes needs to know before index-exact agreement can be
    demanded: ``iou_margin`` = min |IoU - thr| over every IoU test performed; ``pair_gap`` = min score gap between
    a kept box and a box it suppressed (a swap of the two would keep the other one); ``kept_gap`` = min score gap
    between consecutive kept boxes (a swap reorders the output)."""
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    n = boxes.shape[0]
    assert np.all(scores[:-1] >= scores[1:]), "candidates must be sorted by score desc"
    suppressed = np.zeros(n, dtype=bool)
    keep: List[int] = []
    iou_margin, pair_gap = np.inf, np.inf
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if max_keep is not None and len(keep) >= max_keep:
            break
        rest = np.nonzero((labels[i + 1:] == labels[i]) & ~suppressed[i + 1:])[0] + i + 1
        if rest.size:
            sup = iou_suppresses(boxes[i], boxes[rest], thr)
            suppressed[rest[sup]] = True
            if stats is not None:
                ovr = _iou_f32(boxes[i], boxes[rest]).astype(np.float64)
                ovr = ovr[np.isfinite(ovr)]
                if ovr.size:
                    iou_margin = min(iou_margin, float(np.min(np.abs(ovr - float(f32(thr))))))
                if np.any(sup):
                    pair_gap = min(pair_gap, float(np.min(scores[i].astype(np.float64) - scores[rest[sup]].astype(np.float64))))
    if stats is not None:
        ks = scores[np.asarray(keep, dtype=np.int64)].astype(np.float64) if keep else np.zeros(0)
        stats["iou_margin"] = iou_margin if np.isfinite(iou_margin) else 1.0
        stats["pair_gap"] = pair_gap if np.isfinite(pair_gap) else 1.0
        stats["kept_gap"] = float(np.min(-np.diff(ks))) if ks.size > 1 else 1.0
    return np.asarray(keep, dtype=np.int64)


def cut_gap(scores: np.ndarray, score_thr: float, topk: int) -> float:
    """Score gap at the ``nms_pre`` cut of filter_scores_and_topk: (last candidate taken) - (first one left out); 1.0
    when every valid candidate fits."""
    flat = np.ascontiguousarray(scores, dtype=f32).reshape(-1)
    valid = flat[flat > f32(score_thr)]
    if valid.shape[0] <= topk:
        return 1.0
    part = np.partition(valid, valid.shape[0] - topk - 1)
    return float(part[valid.shape[0] - topk:].min().astype(np.float64) - part[valid.shape[0] - topk - 1].astype(np.float64))


def unletterbox(boxes: np.ndarray, pad_xy: Tuple[float, float], ratio: float,
                ori_hw: Tuple[int, int], rescale: bool = True) -> np.ndarray:
    """generate_proposal.py:1106-1115: subtract (dw, dh), divide by ratio, clamp."""
    b = boxes.astype(f32).copy()
    b -= np.asarray([pad_xy[0], pad_xy[1], pad_xy[0], pad_xy[1]], dtype=f32)
    if rescale:
        b /= f32(ratio)
    b[:, 0::2] = np.clip(b[:, 0::2], f32(0), f32(ori_hw[1]))
    b[:, 1::2] = np.clip(b[:, 1::2], f32(0), f32(ori_hw[0]))
    return b


def uni_predict_image(boxes: np.ndarray, embed: np.ndarray, scores: np.ndarray,
                      level_of: np.ndarray, logit_scale: np.ndarray, contrast_bias: np.ndarray,
                      num_proposals: int = 300, nms_pre: int = 30000, iou_thr: float = 0.7,
                      score_thr: float = 0.0) -> Dict[str, np.ndarray]:
    """One image of SimpleYOLOWorldDetector.head_predict (generate_proposal.py:1197-1217,
    with the extra outputs of extract_embedding.py:1247-1259).  Boxes stay in
    letterboxed network coordinates (NMS runs before the un-letterbox there)."""
    s, labels, anchors = filter_scores_and_topk(scores, score_thr, nms_pre)
    cand_boxes = boxes[anchors]
    stats: dict = {}
    keep = batched_nms(cand_boxes, s, labels, iou_thr, max_keep=num_proposals, stats=stats)
    stats["cut_gap"] = cut_gap(scores, score_thr, nms_pre)
    a = anchors[keep]
    lv = level_of[a]
    return dict(bboxes=cand_boxes[keep], embeddings=embed[a], scores=s[keep], labels=labels[keep],
                anchors=a, scales=logit_scale[lv].astype(f32), bias=contrast_bias[lv].astype(f32),
                num_candidates=np.int64(s.shape[0]), keep=keep, margins=stats)


def rescale_boxes(b: np.ndarray, pad_xy, scale_xy) -> np.ndarray:
    """(box - [px, py, px, py]) / [sx, sy, sx, sy] in fp32 (yolo_world_head.py:728-734;
    generate_proposal.py:1108-1113 with sx = sy = ratio)."""
    b = b.astype(f32) - np.asarray([pad_xy[0], pad_xy[1], pad_xy[0], pad_xy[1]], dtype=f32)
    return b / np.asarray([scale_xy[0], scale_xy[1], scale_xy[0], scale_xy[1]], dtype=f32)


def clamp_boxes(b: np.ndarray, ori_hw) -> np.ndarray:
    b = b.astype(f32).copy()
    b[:, 0::2] = np.clip(b[:, 0::2], f32(0), f32(ori_hw[1]))
    b[:, 1::2] = np.clip(b[:, 1::2], f32(0), f32(ori_hw[0]))
    return b


def mmdet_predict_image_from_candidates(cand_boxes, s, labels, pad_param, scale_factor, ori_hw,
                                        iou_thr: float = 0.7, max_per_img: int = 300):
    """Tail of predict_by_feat once candidates exist (yolo_world_head.py:724-746): rescale
    to original pixels, THEN NMS, [:max_per_img], clamp.  pad_param = (top, bottom, left,
    right); scale_factor = (w, h)."""
    b = cand_boxes.astype(f32)
    pad = (0.0, 0.0) if pad_param is None else (pad_param[2], pad_param[0])
    b = rescale_boxes(b, pad, scale_factor)
    stats: dict = {}
    keep = batched_nms(b, s, labels, iou_thr, max_keep=max_per_img, stats=stats)
    return dict(bboxes=clamp_boxes(b[keep], ori_hw), keep=keep, margins=stats)


def mmdet_predict_image(boxes: np.ndarray, scores: np.ndarray, pad_param, scale_factor,
                        ori_hw: Tuple[int, int], score_thr: float = 0.001, nms_pre: int = 30000,
                        iou_thr: float = 0.7, max_per_img: int = 300) -> Dict[str, np.ndarray]:
    """One image of YOLOWorldHead.predict_by_feat, multi_label=True
    (yolo_world_head.py:680-748): filter/top-k, rescale to original pixels, THEN NMS,
    [:max_per_img], clamp."""
    s, labels, anchors = filter_scores_and_topk(scores, score_thr, nms_pre)
    if s.shape[0] == 0:
        return dict(bboxes=np.zeros((0, 4), f32), scores=s, labels=labels, anchors=anchors)
    r = mmdet_predict_image_from_candidates(boxes[anchors], s, labels, pad_param, scale_factor, ori_hw,
                                            iou_thr, max_per_img)
    keep = r["keep"]
    r["margins"]["cut_gap"] = cut_gap(scores, score_thr, nms_pre)
    return dict(bboxes=r["bboxes"], scores=s[keep], labels=labels[keep], anchors=anchors[keep], margins=r["margins"])


def retrieval_scores(embedding: np.ndarray, text: np.ndarray, scale: np.ndarray,
                     bias: np.ndarray) -> np.ndarray:
    """retrieval_metric.py:369-375: sigmoid((E Tᵀ)·exp(scale_r)+bias_r), max over regions.
    embedding [R,768], text [K,768], scale/bias [R] -> [K] fp32 (zeros-size R -> -inf guard
    is not in the reference: R >= 1 always there)."""
    import torch  # the reference computes this in torch fp32; keep its op sequence
    e = torch.from_numpy(np.ascontiguousarray(embedding, dtype=f32))
    t = torch.from_numpy(np.ascontiguousarray(text, dtype=f32))
    sc = torch.from_numpy(np.ascontiguousarray(scale, dtype=f32))
    bs = torch.from_numpy(np.ascontiguousarray(bias, dtype=f32))
    lg = torch.einsum('bw,kw->bk', e, t)
    lg = torch.sigmoid(lg * sc.exp().unsqueeze(1) + bs.unsqueeze(1))
    return torch.max(lg, dim=0)[0].numpy()


def shard_indices(total: int, world: int, rank: int) -> range:
    """InferenceSampler._get_local_indices (extract_embedding.py:1631-1638)."""
    shard = total // world
    left = total % world
    sizes = [shard + int(r < left) for r in range(world)]
    begin = sum(sizes[:rank])
    end = min(sum(sizes[:rank + 1]), total)
    return range(begin, end)
