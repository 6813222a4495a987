"""Offline evaluators.

* ``eval_recalls`` — class-agnostic proposal recall (AR@k), eval_recall/recall.py:118-178: per image the
  IoU matrix ground truths x top-k proposals and the greedy one-to-one assignment run on the device
  (``wd_recall_match``); sorting proposals by score and turning the matched IoUs into recalls stay on the
  host exactly as the reference writes them (numpy), so the result is bit-identical.
* ``evaluate_retrieval_per_class`` — per-class precision / recall / F1 of retrieved image ids,
  eval_retrieval/retrieval_metric.py:14-47 (set arithmetic on Python ints: host code).
* the retrieval hand-over file — ``retrieval_records`` / ``save_retrieval_file`` write exactly what
  extract_embedding.py:1763-1774 saves (``{"image_embedding": [{image_id, embedding, scale, bias}, ...],
  "text_embedding": [K, 768]}`` through ``torch.save``), so the reference's ``retrieval_metric.py`` reads files
  written here and ``retrieval_predictions`` reads files the reference wrote; the latter is
  retrieval_metric.py:362-377 with the scoring (sigmoid(E T^T exp(scale) + bias), max over regions) on the device
  (``wd_retrieval_max``), and ``macro_average`` is the summary at 389-391."""
from __future__ import annotations

from collections.abc import Sequence
from typing import Dict, List, Sequence as Seq, Set, Tuple

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
                 use_legacy_coordinate: bool = False, device="cuda", scratch_budget: int = 64 << 20) -> np.ndarray:
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
    out = torch.zeros(nb, total_gt, dtype=torch.float32, device=dev)
    # The kernel's scratch is (largest gt count x largest proposal count of the launch) per (image, budget) block.  Images
    # are therefore processed in chunks: consecutive images are added while chunk-maximum x images x budgets stays under
    # scratch_budget floats (COCO val in one piece would take 1.6 GB, LVIS-scale sets tens of GB).
    cap_p = int(proposal_nums.max())
    i0 = 0
    while i0 < n_img:
        i1, mg, mp = i0, 1, 1
        while i1 < n_img:
            mg2 = max(mg, g_list[i1].shape[0], 1)
            mp2 = max(mp, min(p_list[i1].shape[0], cap_p), 1)
            need = int(L.LIB.wd_recall_scratch_floats(mg2, mp2)) * (i1 + 1 - i0) * nb
            if i1 > i0 and need > scratch_budget:
                break
            mg, mp, i1 = mg2, mp2, i1 + 1
        per_block = max(int(L.LIB.wd_recall_scratch_floats(mg, mp)), 1)
        n_c = i1 - i0
        scratch = torch.empty(per_block * n_c * nb, dtype=torch.float32, device=dev)
        g_loc = torch.from_numpy((g_off[i0:i1 + 1] - g_off[i0]).astype(np.int32)).to(dev)   # keep alive across the launch
        p_loc = torch.from_numpy((p_off[i0:i1 + 1] - p_off[i0]).astype(np.int32)).to(dev)
        if int(g_off[i1]) > int(g_off[i0]):
            L.check(L.LIB.wd_recall_match(gt_d.data_ptr() + int(g_off[i0]) * 16, g_loc.data_ptr(),
                                          pr_d.data_ptr() + int(p_off[i0]) * 16, p_loc.data_ptr(), n_c, budgets.data_ptr(), nb,
                                          scratch.data_ptr(), per_block, out.data_ptr() + int(g_off[i0]) * 4, total_gt,
                                          int(bool(use_legacy_coordinate)), L.stream_ptr()), "wd_recall_match")
        i0 = i1
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


# ------------------------------------------------------------------------------------------ retrieval hand-over file
def retrieval_records(image_ids: Seq[int], embeddings: torch.Tensor, counts: torch.Tensor, scales: torch.Tensor,
                      bias: torch.Tensor) -> List[dict]:
    """Fixed-shape detector / gather outputs -> the per-image records of extract_embedding.py:1765-1773:
    ``embeddings`` [N, R, D], ``scales`` / ``bias`` [N, R], ``counts`` [N] (kept regions per image); every record holds
    CPU tensors trimmed to its count, ``image_id`` a Python int."""
    e, sc, bi = embeddings.detach().cpu(), scales.detach().cpu(), bias.detach().cpu()
    cn = [int(v) for v in counts.detach().cpu().tolist()]
    if not (len(image_ids) == e.shape[0] == sc.shape[0] == bi.shape[0] == len(cn)):
        raise ValueError("image_ids, embeddings, counts, scales and bias must describe the same images")
    if any(c < 0 or c > e.shape[1] for c in cn):
        raise ValueError("count outside [0, regions per image]")
    return [{"image_id": int(i), "embedding": e[n, :c].clone(), "scale": sc[n, :c].clone(), "bias": bi[n, :c].clone()}
            for n, (i, c) in enumerate(zip(image_ids, cn))]


def save_retrieval_file(path: str, records: List[dict], text_embeddings: torch.Tensor) -> None:
    """extract_embedding.py:1774."""
    torch.save({"image_embedding": records, "text_embedding": text_embeddings.detach().cpu()}, path)


def load_retrieval_file(path: str) -> dict:
    """retrieval_metric.py:362."""
    pred = torch.load(path, map_location="cpu")
    if not isinstance(pred, dict) or "image_embedding" not in pred or "text_embedding" not in pred:
        raise ValueError(f"{path} is not a retrieval file (needs image_embedding and text_embedding)")
    return pred


def retrieval_scores(records: List[dict], text_embedding: torch.Tensor, device="cuda", images_per_launch: int = 256,
                     precision=None) -> torch.Tensor:
    """[len(records), K] fp32 on the device: per image the max over its regions of
    sigmoid(<e, t_k> * exp(scale) + bias) (retrieval_metric.py:369-375); images without regions score 0."""
    dev = torch.device(device)
    t = text_embedding.to(dev, torch.float32).contiguous()
    k, dim = t.shape
    out = torch.zeros(len(records), k, dtype=torch.float32, device=dev)
    from .parallel import BankScorer
    scorer = BankScorer(t, precision)          # fp16x3 under its range guard by default (round 5; fp32 before)
    for lo in range(0, len(records), images_per_launch):
        chunk = records[lo:lo + images_per_launch]
        rows = max(1, max(int(r["embedding"].shape[0]) for r in chunk))
        e = torch.zeros(len(chunk), rows, dim, dtype=torch.float32)
        sc, bi = torch.zeros(len(chunk), rows, dtype=torch.float32), torch.zeros(len(chunk), rows, dtype=torch.float32)
        cnt = torch.zeros(len(chunk), dtype=torch.int32)
        for i, r in enumerate(chunk):
            n = int(r["embedding"].shape[0])
            if r["embedding"].shape[1:] != (dim,) or r["scale"].shape[0] != n or r["bias"].shape[0] != n:
                raise ValueError(f"record of image {r.get('image_id')} is inconsistent with the text bank / itself")
            e[i, :n], sc[i, :n], bi[i, :n], cnt[i] = r["embedding"], r["scale"], r["bias"], n
        ed, sd, bd, cd = e.to(dev), sc.to(dev), bi.to(dev), cnt.to(dev)
        scorer(ed, cd, sd, bd, out=out[lo:lo + len(chunk)])
    return out


def retrieval_predictions(pred: dict, classnames: Seq[str], thre: float, device="cuda") -> Dict[str, List[int]]:
    """retrieval_metric.py:365-377: class name -> ids of the images whose best region scores above ``thre``."""
    if len(classnames) != pred["text_embedding"].shape[0]:
        raise ValueError("one class name per text embedding row")
    records = pred["image_embedding"]
    scores = retrieval_scores(records, pred["text_embedding"], device).cpu()
    out: Dict[str, List[int]] = {name: [] for name in classnames}
    for r, row in zip(records, scores):
        for ids in torch.where(row > thre)[0].tolist():
            out[classnames[ids]].append(r["image_id"])
    return out


def macro_average(results: Dict[str, Dict[str, float]]) -> Tuple[float, float, float]:
    """retrieval_metric.py:389-391."""
    vals = list(results.values())
    return (float(np.mean([r["precision"] for r in vals])), float(np.mean([r["recall"] for r in vals])),
            float(np.mean([r["f1"] for r in vals])))
