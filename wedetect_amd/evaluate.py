"""Offline evaluators.

* ``eval_recalls`` — class-agnostic proposal recall (AR@k), eval_recall/recall.py:118-178: per image the
  IoU matrix ground truths x top-k proposals and the greedy one-to-one assignment run on the device
  (``wd_recall_match``); sorting proposals by score and turning the matched IoUs into recalls stay on the
  host exactly as the reference writes them (numpy), so the result is bit-identical.
* ``evaluate_retrieval_per_class`` — per-class precision / recall / F1 of retrieved image ids,
  eval_retrieval/retrieval_metric.py:14-47 (set arithmetic on Python ints: host code)."""
from __future__ import annotations

from collections.abc import Sequence
from typing import Dict, List, Set

import numpy as np
import torch

from . import lib as L


def set_recall_param(proposal_nums, iou_thrs):
    """recall.py:103-121."""
    if isinstance(proposal_nums, Sequence):
        _p = np.array(proposal_nums)
    elif isinstance(proposal_nums, int):
        _p = np.array([proposal_nums])
    else:
        _p = proposal_nums
    if iou_thrs is None:
        _t = np.array([0.5])
    elif isinstance(iou_thrs, Sequence):
        _t = np.array(iou_thrs)
    elif isinstance(iou_thrs, float):
        _t = np.array([iou_thrs])
    else:
        _t = iou_thrs
    return _p, _t


def matched_ious(gts: List[np.ndarray], proposals: List[np.ndarray], proposal_nums: np.ndarray,
                 use_legacy_coordinate: bool = False, device="cuda") -> np.ndarray:
    """[len(proposal_nums), total_gt] fp32: the ``_ious`` array of recall.py:70-92 before sorting."""
    n_img = len(gts)
    g_list = [np.zeros((0, 4), np.float32) if g is None else np.asarray(g, np.float32).reshape(-1, 4) for g in gts]
    p_list = [np.asarray(p, np.float32)[:, :4].reshape(-1, 4) for p in proposals]
    g_off = np.zeros(n_img + 1, np.int32)
    p_off = np.zeros(n_img + 1, np.int32)
    g_off[1:] = np.cumsum([g.shape[0] for g in g_list])
    p_off[1:] = np.cumsum([p.shape[0] for p in p_list])
    total_gt = int(g_off[-1])
    nb = int(proposal_nums.size)
    if total_gt == 0:
        return np.zeros((nb, 0), np.float32)
    max_gt = max(g.shape[0] for g in g_list)
    max_p = max(max(p.shape[0] for p in p_list), 1)
    if max_gt >= 65535 or max_p >= 65535:
        raise L.WedetectHipError("wd_recall_match handles fewer than 65535 boxes per image")
    dev = torch.device(device)
    cat = lambda xs: torch.from_numpy(np.ascontiguousarray(np.concatenate(xs, 0) if xs else np.zeros((0, 4), np.float32)))
    gt_d = cat(g_list).to(dev)
    pr_d = (cat(p_list) if int(p_off[-1]) else torch.zeros(1, 4)).to(dev)
    budgets = torch.from_numpy(proposal_nums.astype(np.int32)).to(dev)
    per_block = int(L.LIB.wd_recall_scratch_floats(max_gt, min(max_p, int(proposal_nums.max()))))
    scratch = torch.empty(max(per_block, 1) * n_img * nb, dtype=torch.float32, device=dev)
    out = torch.zeros(nb, total_gt, dtype=torch.float32, device=dev)
    g_off_d, p_off_d = torch.from_numpy(g_off).to(dev), torch.from_numpy(p_off).to(dev)      # keep alive across the launch
    L.check(L.LIB.wd_recall_match(gt_d.data_ptr(), g_off_d.data_ptr(), pr_d.data_ptr(),
                                  p_off_d.data_ptr(), n_img, budgets.data_ptr(), nb,
                                  scratch.data_ptr(), max(per_block, 1), out.data_ptr(), total_gt,
                                  int(bool(use_legacy_coordinate)), L.stream_ptr()), "wd_recall_match")
    return out.cpu().numpy()


def eval_recalls(gts, proposals, proposal_nums=None, iou_thrs=0.5, use_legacy_coordinate=False, device="cuda"):
    """recall.py:118-178 (without the table print): recalls [len(proposal_nums), len(iou_thrs)]."""
    img_num = len(gts)
    assert img_num == len(proposals)
    proposal_nums, iou_thrs = set_recall_param(proposal_nums, iou_thrs)
    props = []
    for i in range(img_num):
        p = np.asarray(proposals[i])
        if p.ndim == 2 and p.shape[1] == 5:
            p = p[np.argsort(p[:, 4])[::-1], :]                 # recall.py:150-153
        props.append(p[: min(p.shape[0], int(proposal_nums[-1]))])
    ious = matched_ious(gts, props, proposal_nums, use_legacy_coordinate, device)
    total_gt_num = ious.shape[1]
    ious = np.fliplr(np.sort(ious, axis=1))
    recalls = np.zeros((proposal_nums.size, iou_thrs.size))
    for i, thr in enumerate(iou_thrs):
        recalls[:, i] = (ious >= thr).sum(axis=1) / float(total_gt_num)
    return recalls


def evaluate_retrieval_per_class(predictions: Dict[str, List[int]], gt: Dict[str, Set[int]]) -> Dict[str, Dict[str, float]]:
    """retrieval_metric.py:14-47."""
    results = {}
    for cat_name in set(gt.keys()):
        pred_set = set(map(int, predictions.get(cat_name, [])))
        gt_set = gt[cat_name]
        if len(gt_set) == 0:
            continue
        tp, fp, fn = len(pred_set & gt_set), len(pred_set - gt_set), len(gt_set - pred_set)
        precision = tp / (tp + fp) if (tp + fp) > 0 else 0.0
        recall = tp / (tp + fn) if (tp + fn) > 0 else 0.0
        f1 = 2 * precision * recall / (precision + recall) if (precision + recall) > 0 else 0.0
        results[cat_name] = {"precision": round(precision, 4), "recall": round(recall, 4), "f1": round(f1, 4),
                             "support": len(gt_set), "n_pred": len(pred_set)}
    return results
