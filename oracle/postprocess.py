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
  batched NMS — third-party native code, absent from /root/reference and from this image
  (**parity unpinned**: no binary to run against), restated from the PUBLISHED sources, branch
  for branch, in the section "third-party NMS" below:
    torchvision.ops.batched_nms   (call generate_proposal.py:1210)    torchvision/ops/boxes.py:
        batched_nms / _batched_nms_coordinate_trick / _batched_nms_vanilla, CPU kernel
        torchvision/csrc/ops/cpu/nms_kernel.cpp: nms_kernel_impl
    mmcv.ops.batched_nms 2.1.0    (call mmdet BaseDenseHead._bbox_post_process, reached from
        yolo_world_head.py:740-744 with config/wedetect_base.py:18-25 nms=dict(type='nms',
        iou_threshold=0.7))       mmcv/ops/nms.py: batched_nms / nms / NMSop.forward, CPU kernel
        mmcv/ops/csrc/pytorch/cpu/nms.cpp: nms_cpu
  Both CPU kernels take the boxes in descending score (their internal sorts are given the
  defined total order: stable), and box i suppresses a later, not yet suppressed box j iff
  inter / (area_i + area_j - inter) > thr, area = (x2-x1)*(y2-y1), inter = max(0,.)*max(0,.),
  fp32, evaluated left to right.  Class awareness is NOT a label test in either library:
  it is the coordinate offset  boxes + label * (boxes.max() + 1)  (fp32: it quantises the
  boxes — 0.008 px at label 79, 0.125 px at label 1202 on 1280-px coordinates) followed by a
  class-agnostic NMS, except that torchvision switches to a per-class loop on the ORIGINAL
  boxes above 4000 box coordinates on the CPU (20000 on a GPU) and mmcv to a per-class loop
  on the OFFSET boxes from split_thr = 10000 candidates.
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
                max_keep: Optional[int] = None, stats: Optional[dict] = None, compare: str = "f32") -> np.ndarray:
    """Label-test form of class-aware greedy NMS == torchvision's _batched_nms_vanilla on sorted input (one global
    loop instead of one per class; tests/test_cpu.py checks the two equal).  It is what WD_NMS_VANILLA computes; the
    reference's call sites go through torchvision_batched_nms / mmcv_batched_nms below.  Inputs must already be in the
    defined total order (they are: the output of filter_scores_and_topk).  Returns kept candidate indices
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
            if compare == "double":          # torchvision: fp32 IoU promoted and compared with the C++ double threshold
                sup = _iou_f32(boxes[i], boxes[rest]).astype(np.float64) > float(thr)
            suppressed[rest[sup]] = True
            if stats is not None:
                rel = rest <= stats.get("rmax", n)           # effective margins: only candidates that can reach the output
                ovr = _iou_f32(boxes[i], boxes[rest[rel]]).astype(np.float64)
                ovr = ovr[np.isfinite(ovr)]
                if ovr.size:
                    iou_margin = min(iou_margin, float(np.min(np.abs(ovr - float(f32(thr))))))
                if np.any(sup & rel):
                    pair_gap = min(pair_gap, float(np.min(scores[i].astype(np.float64) - scores[rest[sup & rel]].astype(np.float64))))
    if stats is not None:
        ks = scores[np.asarray(keep, dtype=np.int64)].astype(np.float64) if keep else np.zeros(0)
        stats["iou_margin"] = iou_margin if np.isfinite(iou_margin) else 1.0
        stats["pair_gap"] = pair_gap if np.isfinite(pair_gap) else 1.0
        stats["kept_gap"] = float(np.min(-np.diff(ks))) if ks.size > 1 else 1.0
    return np.asarray(keep, dtype=np.int64)


# ------------------------------------------------------------------------------------------ third-party NMS
TV_TRICK_MAX_NUMEL = {"cpu": 4000, "cuda": 20000}     # torchvision/ops/boxes.py: batched_nms
MMCV_SPLIT_THR = 10000                                   # mmcv/ops/nms.py: batched_nms, nms_cfg default


def _merge_stats(stats: Optional[dict], iou_margin: float, pair_gap: float) -> None:
    if stats is None:
        return
    stats["iou_margin"] = min(stats.get("iou_margin", 1.0), iou_margin)
    stats["pair_gap"] = min(stats.get("pair_gap", 1.0), pair_gap)


def nms_greedy(boxes: np.ndarray, scores: np.ndarray, thr: float, compare: str = "f32",
               stats: Optional[dict] = None, max_keep: Optional[int] = None) -> np.ndarray:
    """Class-AGNOSTIC greedy NMS, the loop both CPU kernels run:
      torchvision/csrc/ops/cpu/nms_kernel.cpp nms_kernel_impl  (``compare="double"``: ``iou_threshold`` is a C++
          double there, the fp32 ``ovr`` is promoted for ``ovr > iou_threshold``; order = stable sort of the scores)
      mmcv/ops/csrc/pytorch/cpu/nms.cpp nms_cpu, offset = 0     (``compare="f32"``: ``float iou_threshold``)
    Returns kept indices into ``boxes`` in descending score (ties: index ascending).  ``max_keep`` stops early — equal
    to slicing the full result, a box's fate depends on earlier kept boxes only."""
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    scores = np.ascontiguousarray(scores, dtype=f32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    order = np.argsort(-scores, kind="stable")
    bs = boxes[order]
    areas = (bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])
    t64, t32 = float(thr), f32(thr)
    suppressed = np.zeros(n, dtype=bool)
    keep: List[int] = []
    iou_margin, pair_gap = 1.0, 1.0
    for _i in range(n):
        if suppressed[_i]:
            continue
        keep.append(_i)
        if max_keep is not None and len(keep) >= max_keep:
            break
        rest = np.nonzero(~suppressed[_i + 1:])[0] + _i + 1
        if not rest.size:
            continue
        bi, bj = bs[_i], bs[rest]
        w = np.maximum(f32(0), np.minimum(bi[2], bj[:, 2]) - np.maximum(bi[0], bj[:, 0]))
        h = np.maximum(f32(0), np.minimum(bi[3], bj[:, 3]) - np.maximum(bi[1], bj[:, 1]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[_i] + areas[rest] - inter)
        sup = (ovr.astype(np.float64) > t64) if compare == "double" else (ovr > t32)
        suppressed[rest[sup]] = True
        if stats is not None:
            rel = rest <= stats.get("rmax", n)               # effective margins (sorted input: position = candidate rank)
            o64 = ovr.astype(np.float64)[rel]
            o64 = o64[np.isfinite(o64)]
            if o64.size:
                iou_margin = min(iou_margin, float(np.min(np.abs(o64 - (t64 if compare == "double" else float(t32))))))
            if np.any(sup & rel):
                ss = scores[order]
                pair_gap = min(pair_gap, float(np.min(ss[_i].astype(np.float64) - ss[rest[sup & rel]].astype(np.float64))))
    _merge_stats(stats, iou_margin, pair_gap)
    return order[np.asarray(keep, dtype=np.int64)]


def coordinate_offsets(boxes: np.ndarray, idxs: np.ndarray) -> np.ndarray:
    """``max_coordinate = boxes.max(); offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes));
    boxes_for_nms = boxes + offsets[:, None]`` — the same three lines in torchvision/ops/boxes.py
    (_batched_nms_coordinate_trick) and mmcv/ops/nms.py (batched_nms).  All fp32."""
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    max_coordinate = boxes.max()
    offsets = idxs.astype(f32) * (max_coordinate + f32(1))
    return boxes + offsets[:, None]


def torchvision_batched_nms(boxes: np.ndarray, scores: np.ndarray, idxs: np.ndarray, iou_threshold: float,
                            device_type: str = "cpu", stats: Optional[dict] = None,
                            max_keep: Optional[int] = None) -> np.ndarray:
    """torchvision.ops.batched_nms (torchvision/ops/boxes.py), the function generate_proposal.py:1210 calls:
    ``if boxes.numel() > (4000 if boxes.device.type == "cpu" else 20000): _batched_nms_vanilla else
    _batched_nms_coordinate_trick``.  ``max_keep`` is the caller's ``[:num_proposals]`` applied early where that
    is exact (the agnostic call); the vanilla branch runs every class to the end, as upstream."""
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    scores = np.ascontiguousarray(scores, dtype=f32)
    if boxes.size > TV_TRICK_MAX_NUMEL[device_type]:
        # _batched_nms_vanilla: nms() per class on the boxes as given
        if max_keep is not None and _sorted_desc(scores):
            # same keeps as the per-class loops + final stable sort below, found by ONE greedy pass with a label test
            # that can stop at max_keep (tests/test_cpu.py checks the two equal): what the CPU baseline times
            return batched_nms(boxes, scores, idxs, iou_threshold, max_keep, stats, compare="double")
        keep_mask = np.zeros(scores.shape[0], dtype=bool)
        for class_id in np.unique(idxs):
            curr_indices = np.nonzero(idxs == class_id)[0]
            curr_keep = nms_greedy(boxes[curr_indices], scores[curr_indices], iou_threshold, "double", stats)
            keep_mask[curr_indices[curr_keep]] = True
        keep_indices = np.nonzero(keep_mask)[0]
        out = keep_indices[np.argsort(-scores[keep_indices], kind="stable")]
        return out if max_keep is None else out[:max_keep]
    # _batched_nms_coordinate_trick
    if boxes.size == 0:
        return np.zeros(0, dtype=np.int64)
    return nms_greedy(coordinate_offsets(boxes, idxs), scores, iou_threshold, "double", stats, max_keep)


def mmcv_batched_nms(boxes: np.ndarray, scores: np.ndarray, idxs: np.ndarray, nms_cfg: dict,
                     stats: Optional[dict] = None, max_keep: Optional[int] = None) -> np.ndarray:
    """mmcv.ops.batched_nms (mmcv/ops/nms.py, 2.1.0) for ``nms_cfg = dict(type='nms', iou_threshold=...)``
    [+ ``split_thr``, ``class_agnostic``]; returns ``keep`` (the ``dets`` it also returns are boxes[keep] | scores[keep]).
    ``nms`` -> NMSop.forward: score_threshold 0, max_num -1, offset 0 -> ext_module.nms = nms_cpu.
    ``max_keep`` is mmdet's ``results[:cfg.max_per_img]`` (BaseDenseHead._bbox_post_process) applied early where exact."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop("class_agnostic", False)
    assert cfg.pop("type", "nms") == "nms"
    split_thr = cfg.pop("split_thr", MMCV_SPLIT_THR)
    thr = cfg.pop("iou_threshold")
    assert not cfg, f"unsupported nms_cfg keys {sorted(cfg)}"
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    scores = np.ascontiguousarray(scores, dtype=f32)
    boxes_for_nms = boxes if class_agnostic else coordinate_offsets(boxes, idxs)
    if boxes_for_nms.shape[0] < split_thr:
        return nms_greedy(boxes_for_nms, scores, thr, "f32", stats, max_keep)
    if max_keep is not None and _sorted_desc(scores):
        # per-class loops on the offset boxes + stable sort + [:max_keep] == one label-test pass over them with early exit
        return batched_nms(boxes_for_nms, scores, idxs, thr, max_keep, stats)
    total_mask = np.zeros(scores.shape[0], dtype=bool)
    for id_ in np.unique(idxs):
        mask = np.nonzero(idxs == id_)[0]
        keep = nms_greedy(boxes_for_nms[mask], scores[mask], thr, "f32", stats)
        total_mask[mask[keep]] = True
    keep = np.nonzero(total_mask)[0]
    keep = keep[np.argsort(-scores[keep], kind="stable")]
    return keep if max_keep is None else keep[:max_keep]


def _sorted_desc(scores: np.ndarray) -> bool:
    return bool(np.all(scores[:-1] >= scores[1:]))


def _finish_stats(stats: dict, scores: np.ndarray, keep: np.ndarray) -> dict:
    ks = scores[keep].astype(np.float64)
    stats.setdefault("iou_margin", 1.0)
    stats.setdefault("pair_gap", 1.0)
    stats["kept_gap"] = float(np.min(-np.diff(ks))) if ks.size > 1 else 1.0
    return stats


def cut_gap(scores: np.ndarray, score_thr: float, topk: int) -> float:
    """Score gap at the ``nms_pre`` cut of filter_scores_and_topk: (last candidate taken) - (first one left out); 1.0
    when every valid candidate fits."""
    flat = np.ascontiguousarray(scores, dtype=f32).reshape(-1)
    valid = flat[flat > f32(score_thr)]
    if valid.shape[0] <= topk:
        return 1.0
    part = np.partition(valid, valid.shape[0] - topk - 1)
    return float(part[valid.shape[0] - topk:].min().astype(np.float64) - part[valid.shape[0] - topk - 1].astype(np.float64))


def effective_margins(nms_fn, s: np.ndarray, n_keep: int, cut: float) -> np.ndarray:
    """Decision margins restricted to what can change the FIRST ``n_keep`` output rows — [iou, pair, kept, cut] like the
    ``margins`` of the predict functions, which take every decision of the whole candidate list (30 000 candidates, millions
    of IoU tests: some decision always sits inside fp32 noise although it can never reach the output).  ``nms_fn(stats,
    max_keep)`` runs the image's batched NMS on its (sorted) candidates.  A decision matters only if it involves a candidate
    ranked at or before the (n_keep + 1)-th keeper — the first row that is NOT output: later candidates cannot enter the
    first n_keep rows whatever happens to them.  iou / pair: over the IoU tests of the first n_keep + 1 keepers against
    candidates up to that rank; kept: the gaps between consecutive keepers including the one to the first row left out
    (a swap there changes membership); cut: the nms_pre gap only if NMS ran out of candidates before n_keep + 1 keepers."""
    keep = nms_fn(None, n_keep + 1)
    exhausted = keep.shape[0] <= n_keep
    stats = {"rmax": int(s.shape[0]) if exhausted else int(keep[n_keep])}
    keep2 = nms_fn(stats, n_keep + 1)
    assert np.array_equal(keep, keep2)
    ks = s[keep].astype(np.float64)
    kept = float(np.min(-np.diff(ks))) if ks.size > 1 else 1.0
    return np.asarray([stats.get("iou_margin", 1.0), stats.get("pair_gap", 1.0), kept, cut if exhausted else 1.0], dtype=np.float64)


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
                      score_thr: float = 0.0, device_type: str = "cpu", effective: bool = False) -> Dict[str, np.ndarray]:
    """One image of SimpleYOLOWorldDetector.head_predict (generate_proposal.py:1197-1217,
    with the extra outputs of extract_embedding.py:1247-1259).  Boxes stay in
    letterboxed network coordinates (NMS runs before the un-letterbox there);
    ``torchvision.ops.batched_nms(bbox, scores, labels, 0.7)[:num_proposals]`` (1210)."""
    s, labels, anchors = filter_scores_and_topk(scores, score_thr, nms_pre)
    cand_boxes = boxes[anchors]
    stats: dict = {}
    keep = torchvision_batched_nms(cand_boxes, s, labels, iou_thr, device_type, stats, max_keep=num_proposals)
    _finish_stats(stats, s, keep)
    stats["cut_gap"] = cut_gap(scores, score_thr, nms_pre)
    a = anchors[keep]
    lv = level_of[a]
    out = dict(bboxes=cand_boxes[keep], embeddings=embed[a], scores=s[keep], labels=labels[keep],
               anchors=a, scales=logit_scale[lv].astype(f32), bias=contrast_bias[lv].astype(f32),
               num_candidates=np.int64(s.shape[0]), keep=keep, margins=stats)
    if effective:
        out["eff_margins"] = effective_margins(
            lambda st, mk: torchvision_batched_nms(cand_boxes, s, labels, iou_thr, device_type, st, max_keep=mk),
            s, num_proposals, stats["cut_gap"])
    return out


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
                                        iou_thr: float = 0.7, max_per_img: int = 300, nms_cfg: Optional[dict] = None):
    """Tail of predict_by_feat once candidates exist (yolo_world_head.py:724-746): rescale
    to original pixels, THEN mmdet's _bbox_post_process (``if with_nms and results.bboxes.numel() > 0:
    batched_nms(bboxes, scores, labels, cfg.nms)``; ``results[:cfg.max_per_img]``), clamp.
    pad_param = (top, bottom, left, right); scale_factor = (w, h)."""
    b = cand_boxes.astype(f32)
    pad = (0.0, 0.0) if pad_param is None else (pad_param[2], pad_param[0])
    b = rescale_boxes(b, pad, scale_factor)
    stats: dict = {}
    cfg = dict(type="nms", iou_threshold=iou_thr) if nms_cfg is None else nms_cfg
    keep = mmcv_batched_nms(b, s, labels, cfg, stats, max_keep=max_per_img) if b.size else np.zeros(0, np.int64)
    _finish_stats(stats, np.asarray(s, dtype=f32), keep)
    return dict(bboxes=clamp_boxes(b[keep], ori_hw), keep=keep, margins=stats)


def mmdet_predict_image(boxes: np.ndarray, scores: np.ndarray, pad_param, scale_factor,
                        ori_hw: Tuple[int, int], score_thr: float = 0.001, nms_pre: int = 30000,
                        iou_thr: float = 0.7, max_per_img: int = 300, nms_cfg: Optional[dict] = None,
                        effective: bool = False) -> Dict[str, np.ndarray]:
    """One image of YOLOWorldHead.predict_by_feat, multi_label=True
    (yolo_world_head.py:680-748): filter/top-k, rescale to original pixels, THEN NMS,
    [:max_per_img], clamp."""
    s, labels, anchors = filter_scores_and_topk(scores, score_thr, nms_pre)
    if s.shape[0] == 0:
        return dict(bboxes=np.zeros((0, 4), f32), scores=s, labels=labels, anchors=anchors)
    r = mmdet_predict_image_from_candidates(boxes[anchors], s, labels, pad_param, scale_factor, ori_hw,
                                            iou_thr, max_per_img, nms_cfg)
    keep = r["keep"]
    r["margins"]["cut_gap"] = cut_gap(scores, score_thr, nms_pre)
    out = dict(bboxes=r["bboxes"], scores=s[keep], labels=labels[keep], anchors=anchors[keep], margins=r["margins"])
    if effective:
        pad = (0.0, 0.0) if pad_param is None else (pad_param[2], pad_param[0])
        b = rescale_boxes(boxes[anchors].astype(f32), pad, scale_factor)
        cfg = dict(type="nms", iou_threshold=iou_thr) if nms_cfg is None else nms_cfg
        out["eff_margins"] = effective_margins(lambda st, mk: mmcv_batched_nms(b, s, labels, cfg, st, max_keep=mk),
                                               s, max_per_img, r["margins"]["cut_gap"])
    return out


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
