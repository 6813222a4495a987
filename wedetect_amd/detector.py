"""Operator surface of the hot path — the classes callers of the reference construct.

Mirrors (names, arguments, return structures, error behaviour):
  * ``SimpleYOLOWorldDetector``  generate_proposal.py:1052-1218 / extract_embedding.py:1088-1262
        model = SimpleYOLOWorldDetector('base', prompt_dim=768, num_prompts=256)
        model.load_state_dict(ckpt, strict=False); model.cuda(); model.eval()
        outputs = model([path_or_PIL, ...])  ->  list of dict(bboxes, embeddings, scores, labels, scales, bias)
  * ``YOLOWorldDetector``        wedetect/models/detectors/yolo_world.py:19-113, built either from the reference's
        config keywords (``MODELS.build(cfg.model)``: mm_neck, num_train_classes, num_test_classes,
        data_preprocessor, backbone, neck, bbox_head, train_cfg, test_cfg) or directly by size name
        model.reparameterize(texts) / model.test_step(dict(inputs=[...], data_samples=[...]))
        -> data samples with ``.pred_instances.{bboxes, scores, labels}`` (infer_wedetect.py:117-131)

Both are ``torch.nn.Module``s without parameters of their own: the weights live in packed device buffers owned by
``ImageTower``; ``load_state_dict`` / ``state_dict`` / ``_load_from_state_dict`` (what mmengine's ``load_checkpoint``
walks) take and return the reference's key layout.  All tensor math is done by libwedetect_hip.so through
``ImageTower``; this file only does host-side bookkeeping.  There is no CPU fallback: running a detector without a HIP
device / the built library raises.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .arch import EMBED_DIM, all_params, get_arch
from .registry import MODELS

_IMG_SIZE = {"tiny": (640, 640), "base": (640, 640), "large": (1280, 1280), "nano": (128, 128)}
MAX_OUT_ROWS = 1024          # wd_nms_gather keeps its kept-list in LDS (include/wedetect_hip.h)


# ------------------------------------------------------------------------------------------
# checkpoint key handling
# ------------------------------------------------------------------------------------------
def from_uni_keys(sd: Dict[str, object]) -> Dict[str, object]:
    """Inverse of the remap at generate_proposal.py:1236-1254: accepts the pure-torch module
    names (``backbone.stages...``, ``bbox_head.cls_preds.L.{0,1,3,4,6}``) and returns mmdet
    checkpoint names; keys already in mmdet form pass through."""
    out = {}
    slot_map = {"0": ("0", "conv"), "1": ("0", "bn"), "3": ("1", "conv"), "4": ("1", "bn")}
    for k, v in sd.items():
        nk = k
        if k.startswith("backbone.") and not k.startswith("backbone.image_model.") and not k.startswith("backbone.text_model."):
            nk = "backbone.image_model.model." + k[len("backbone."):]
        elif k.startswith("bbox_head.") and not k.startswith("bbox_head.head_module."):
            parts = k.split(".")
            if parts[1] in ("cls_preds", "reg_preds"):
                slot = parts[3]
                if slot == "6":
                    parts[3] = "2"
                elif slot in slot_map:
                    parts[3:4] = list(slot_map[slot])
            nk = "bbox_head.head_module." + ".".join(parts[1:])
        out[nk] = v
    return out


def _to_numpy_sd(sd) -> Dict[str, np.ndarray]:
    out = {}
    for k, v in sd.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().float().numpy()
        out[k] = np.asarray(v)
    return out


class _IncompatibleKeys:
    def __init__(self, missing, unexpected):
        self.missing_keys, self.unexpected_keys = list(missing), list(unexpected)

    def __repr__(self):
        if not self.missing_keys and not self.unexpected_keys:
            return "<All keys matched successfully>"
        return f"_IncompatibleKeys(missing_keys={self.missing_keys}, unexpected_keys={self.unexpected_keys})"


# ------------------------------------------------------------------------------------------
# host-side letterbox (the reference's arithmetic with PIL; the detectors use the device version,
# wedetect_amd/preprocess.py, which is bit-exact with it)
# ------------------------------------------------------------------------------------------
def letterbox(img, new_shape=(640, 640), color=(114, 114, 114)):
    """Keep-ratio resize + centred pad to ``new_shape`` (h, w) like generate_proposal.py:17-82:
    r = min(W'/w, H'/h); new size = round(w*r), round(h*r); bilinear; paste at (dw//2, dh//2);
    returns (PIL image, ratio, (dw/2, dh/2)) — note the float half-pads the reference
    subtracts later (1108-1110) next to the integer paste offset."""
    from PIL import Image
    w, h = img.size
    tw, th = new_shape[1], new_shape[0]
    r = min(tw / w, th / h)
    nw, nh = int(round(w * r)), int(round(h * r))
    resized = img.resize((nw, nh), Image.Resampling.BILINEAR)
    dw, dh = tw - nw, th - nh
    canvas = Image.new("RGB", (tw, th), color)
    canvas.paste(resized, (dw // 2, dh // 2))
    return canvas, r, (dw / 2, dh / 2)


class _TowerHolder:
    """Shared plumbing: weights -> packed device tensors -> one ImageTower per (batch, H, W)."""

    def __init__(self, arch: str, num_prompts: int, max_classes: int, max_out: int, precision: Optional[str] = None,
                 nms_pre: int = 30000):
        if max_out > MAX_OUT_ROWS:
            raise NotImplementedError(f"max_per_img / num_proposals {max_out} > {MAX_OUT_ROWS} (the NMS kernel's kept-list capacity)")
        if nms_pre < 1:
            raise ValueError("nms_pre must be positive")
        self.arch = get_arch(arch)
        self.precision = precision
        self._asked_precision = precision          # what the caller configured: restored when a fallen-back tower returns to fp16x3
        self.num_prompts = num_prompts
        self.max_classes, self.max_out, self.nms_pre = max_classes, max_out, nms_pre
        self._sd: Optional[Dict[str, np.ndarray]] = None
        self._packed = None
        self._towers = {}
        self._graphs = {}
        # batches up to this size replay the step from a captured hipGraph; 0 = always eager.  Default 0 since round 5: with the
        # neck / head issued as a DAG on side streams the eager step is as fast as the replay or faster at every small batch
        # (Tiny batch 1: 3.93 ms eager against 4.35 replayed, Base batch 1: 8.18 against 8.25 — profiles/r05_small_batch.txt; the
        # replay was worth 6.6 -> 5.4 ms in round 2, before the step's launch count and host cost came down)
        self.graph_max_batch = int(os.environ.get("WEDETECT_GRAPH_MAX_BATCH", "0"))
        # fp16x3 split scales are chosen from the first batch a tower sees (engine.ImageTower.calibrate); "0": never
        self.auto_calibrate = os.environ.get("WEDETECT_CALIBRATE", "1") != "0"
        # ONE set of split scales per checkpoint, shared by every (B, H, W) tower: an image gets the same bits whatever batch
        # shape it arrives in (per-tower first-batch calibration broke that for checkpoints with scales != 1)
        self.sscale = None
        self.device = None

    def load(self, sd, strict: bool):
        sd = _to_numpy_sd(from_uni_keys(sd))
        want = {name: shape for name, shape, _ in all_params(self.arch, self.num_prompts)}
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want and not k.endswith("num_batches_tracked")
                      and not k.startswith("backbone.text_model.")
                      and not k.startswith(("backbone.image_model.model.norm.", "backbone.image_model.model.head."))]
        for k, shape in want.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shape)}")
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        if missing:
            raise RuntimeError(f"cannot run with missing tensors (no random-init fallback): {missing[:8]} ...")
        self._sd = {k: sd[k] for k in want}
        self._packed = None
        self._towers.clear()
        self._graphs.clear()
        self.sscale = None                                   # a new checkpoint calibrates anew
        return _IncompatibleKeys(missing, unexpected)

    def state(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in (self._sd or {}).items())

    def tower(self, batch: int, height: int, width: int):
        from .engine import ImageTower
        from .pack import pack
        if self._sd is None:
            raise RuntimeError("load_state_dict() must be called before inference")
        if self.device is None:
            raise RuntimeError("model is not on a HIP device: call .cuda() (there is no CPU execution path)")
        if self._packed is None:
            self._packed = pack(self._sd, self.arch, self.device)
        key = (batch, height, width)
        if key not in self._towers:
            if len(self._towers) >= 4:                       # each tower owns its activation buffers
                old = self._towers.pop(next(iter(self._towers)))
                self._graphs = {k: g for k, g in self._graphs.items() if g.tower is not old}
            self._towers[key] = ImageTower(self.arch, self._packed, batch, height, width, device=self.device,
                                           max_classes=self.max_classes, max_out=self.max_out, nms_pre=self.nms_pre,
                                           precision=self.precision)
            if self.sscale is not None:
                self._towers[key].adopt_scales(self.sscale)
        return self._towers[key]

    def _share_scales(self, tower) -> None:
        self.sscale = dict(tower.sscale)
        for t in self._towers.values():
            if t is not tower:
                t.adopt_scales(self.sscale)

    def recalibrate(self, tower, images_u8) -> None:
        """The range guard tripped under scales taken from an earlier batch: derive them again from this one (scales only go
        down) and hand them to every tower of the checkpoint."""
        tower.calibrate(images_u8, merge=True)
        self._share_scales(tower)

    def detect(self, tower, images_u8, text, meta, **kw):
        """``tower.detect`` for the detector classes.  Small batches (the reference runs batch 1 everywhere) are
        launch-bound, so the step is replayed from a hipGraph captured on first use (engine.GraphedDetect: inputs are
        copied into the graph's static buffers, results are the tower's usual buffers — bit-identical to the eager
        step, tests/test_gpu_detector.py).  A graph belongs to one (tower, arithmetic mode, bank size, thresholds); the
        range guard's switch to fp32 therefore captures anew."""
        if not tower.calibrated and self.auto_calibrate:
            # first batch of this CHECKPOINT: one fp32 pass chooses the fp16x3 split scales for its activation ranges
            # (engine.ImageTower.calibrate; scale 1 everywhere for ordinary checkpoints); towers of other shapes adopt them
            if self.sscale is None:
                tower.calibrate(images_u8)
                self._share_scales(tower)
            else:
                tower.adopt_scales(self.sscale)
        if tower.B > self.graph_max_batch:
            return tower.detect(images_u8, text, meta, **kw)
        k = int(text.shape[0])
        if k > tower.max_classes or k > tower.text_norm.shape[0]:
            # a larger bank re-allocates the tower's similarity / top-k buffers: do it BEFORE any capture or replay, and
            # drop every graph of this tower that still points at the old ones (a K=80 graph replayed after a
            # K=100 call would write into freed memory)
            tower._alloc_post(max(k, tower.max_classes))
            if k > tower.text_norm.shape[0]:
                tower.text_norm = torch.empty(k, tower.text_norm.shape[1], dtype=torch.float32, device=tower.dev)
                tower.generation += 1
        self._graphs = {kk: g for kk, g in self._graphs.items() if g.tower is not tower or g.generation == tower.generation}
        key = (id(tower), tower.generation, tower.precision, tower.neck_pin, k, tuple(sorted((a, str(b)) for a, b in kw.items())))
        g = self._graphs.get(key)
        if g is None:
            from .engine import GraphedDetect
            if len(self._graphs) >= 8:
                self._graphs.pop(next(iter(self._graphs)))
            g = self._graphs[key] = GraphedDetect(tower, k, **kw)
            if g.generation != tower.generation:             # cannot happen after the pre-growth above; never replay a stale graph
                raise RuntimeError("tower buffers were re-allocated during graph capture")
        return g(images_u8, text, meta)


class _DeviceModule(torch.nn.Module):
    """nn.Module plumbing shared by the two detectors: no parameters, weights in ``self._h`` (a _TowerHolder)."""

    _h: _TowerHolder

    def cuda(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: wedetect_amd has no CPU execution path")
        if isinstance(device, (str, torch.device)):
            device = torch.device(device).index
        self._h.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._on_device(self._h.device)
        return self

    def _on_device(self, device) -> None:
        pass

    def to(self, *args, **kwargs):
        device, dtype, _, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype is not None and dtype != torch.float32:
            raise NotImplementedError("the device path computes in fp32 / fp16x3 (ImageTower precision=...); "
                                      f"casting the module to {dtype} is not supported")
        if device is not None:
            if device.type != "cuda":
                raise RuntimeError(f"cannot move to {device}: wedetect_amd has no CPU execution path")
            return self.cuda(device)
        return self

    def half(self):
        return self.to(torch.float16)

    def bfloat16(self):
        return self.to(torch.bfloat16)

    def cpu(self):
        raise RuntimeError("wedetect_amd has no CPU execution path")

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = OrderedDict() if destination is None else destination
        for k, v in self._h.state().items():
            out[prefix + k] = v
        return out

    def _load_state(self, state_dict, strict: bool):
        return self._h.load(state_dict, strict)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        if "state_dict" in state_dict and isinstance(state_dict["state_dict"], dict):
            state_dict = state_dict["state_dict"]           # mmengine checkpoint wrapper
        return self._load_state(state_dict, strict)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """The hook torch's and mmengine's recursive loaders call per module: this module takes every key under
        its prefix (there are no child modules holding parameters)."""
        sub = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        try:
            msg = self._load_state(sub, False)
        except RuntimeError as e:
            error_msgs.append(str(e))
            return
        missing_keys.extend(prefix + k for k in msg.missing_keys)
        if strict:
            unexpected_keys.extend(prefix + k for k in msg.unexpected_keys)


# ------------------------------------------------------------------------------------------
# WeDetect-Uni: proposals + embeddings
# ------------------------------------------------------------------------------------------
class SimpleYOLOWorldDetector(_DeviceModule):
    """Drop-in for generate_proposal.py:1052 / extract_embedding.py:1088."""

    def __init__(self, backbone_size, prompt_dim=768, num_prompts=512, num_proposals=300, img_size=None,
                 precision: Optional[str] = None):
        super().__init__()
        if prompt_dim != EMBED_DIM:
            raise ValueError("prompt_dim must be 768")
        self.backbone_size = backbone_size
        self.num_proposals = num_proposals
        self.img_size = tuple(img_size) if img_size is not None else _IMG_SIZE[backbone_size]
        self._h = _TowerHolder(backbone_size, num_prompts, max(num_prompts, 1), num_proposals, precision)
        self._lb = None
        from .lib import TV_TRICK_MAX_NUMEL
        # which execution of the reference the torchvision NMS reproduces: "cpu" (default — the CPU path BASELINE.json's
        # parity contract names: coordinate trick up to 4000 box coordinates, IoU compared with a C++ double threshold) or
        # "cuda" (what the script's hard-coded .cuda() run takes: 20000, float threshold)
        self.tv_nms_device = os.environ.get("WEDETECT_TV_NMS_DEVICE", "cpu")
        if self.tv_nms_device not in TV_TRICK_MAX_NUMEL:
            raise ValueError(f"WEDETECT_TV_NMS_DEVICE must be one of {sorted(TV_TRICK_MAX_NUMEL)}, not {self.tv_nms_device!r}")
        self.tv_trick_max_numel = TV_TRICK_MAX_NUMEL[self.tv_nms_device]

    @torch.no_grad()
    def forward(self, image_paths: Sequence[Union[str, object]], rescale=True) -> List[Dict[str, torch.Tensor]]:
        res, counts, tower = self.forward_batch(image_paths, rescale)
        out = []
        ls = torch.tensor(tower.lvl_logit_scale, dtype=torch.float32, device=self._h.device)
        cb = torch.tensor(tower.lvl_bias, dtype=torch.float32, device=self._h.device)
        for i, n in enumerate(counts):
            lvl = tower.level_of(res["anchors"][i, :n])
            out.append({
                "bboxes": res["bboxes"][i, :n].clone(),
                "embeddings": res["embeddings"][i, :n].clone(),
                "scores": res["scores"][i, :n].clone(),
                "labels": res["labels"][i, :n].to(torch.int64),
                "scales": ls[lvl],
                "bias": cb[lvl],
            })
        return out

    @torch.no_grad()
    def forward_batch(self, image_paths: Sequence[Union[str, object]], rescale=True):
        """The same step with the results left in the tower's fixed-shape device buffers ([B, R, ...] + counts):
        what the multi-GPU gather exchanges (parallel.gather_results).  Returns (buffers, counts list, tower)."""
        from PIL import Image
        from .preprocess import DeviceLetterbox
        if self._h.device is None:
            raise RuntimeError("model is not on a HIP device: call .cuda() (there is no CPU execution path)")
        imgs, metas = [], []
        for p in image_paths:
            imgs.append(Image.open(p).convert("RGB") if isinstance(p, str) else p)
        # letterbox on the device (wd_letterbox_u8, bit-exact with the PIL resize + paste of the
        # reference): the host only decodes and uploads
        if self._lb is None:
            self._lb = DeviceLetterbox(self.img_size, device=self._h.device)
        x, ratios, pads = self._lb(imgs)
        for img, ratio, (dw, dh) in zip(imgs, ratios, pads):
            w, h = img.size if hasattr(img, "size") and not isinstance(img, (np.ndarray, torch.Tensor)) else (img.shape[1], img.shape[0])
            sc = ratio if rescale else 1.0
            metas.append([dw, dh, 0.0, sc, sc, float(w), float(h), 0.0])
        tower = self._h.tower(len(imgs), self.img_size[0], self.img_size[1])
        meta = torch.tensor(metas, dtype=torch.float32, device=self._h.device)
        # torchvision.ops.batched_nms(bbox, scores, labels, 0.7)[:num_proposals] (generate_proposal.py:1210), with the
        # branch threshold of the device the reference's tensors would live on ($WEDETECT_TV_NMS_DEVICE: "cpu" = the CPU
        # reference path the parity contract names, "cuda" = what the script's hard-coded .cuda() run takes)
        run = lambda: self._h.detect(tower, x, tower.P["prompts"], meta, normalize_text=False, score_thr=0.0, with_embed=True,
                                     nms="torchvision", nms_param=self.tv_trick_max_numel, nms_device=self.tv_nms_device)
        res = run()
        recal = (lambda: self._h.recalibrate(tower, x)) if self._h.auto_calibrate else None
        counts = tower.checked_counts(res, run, recal)      # one D2H sync per batch (+ the fp16x3 range guard)
        # towers built later for other shapes start in fp32 too while this one is in its fallback (round 6: not for good — after
        # ImageTower.FALLBACK_RETRY clean batches it re-calibrates and returns to fp16x3, and so does the holder's default)
        self._h.precision = "fp32" if tower.overflowed else self._h._asked_precision
        return res, counts, tower


# ------------------------------------------------------------------------------------------
# WeDetect: open-vocabulary detection against a text bank
# ------------------------------------------------------------------------------------------
class InstanceData:
    """Stand-in for mmengine.structures.InstanceData as the reference's scripts and mmdet's result handling use
    it: attribute and item access, boolean-mask / index / slice ``__getitem__``, ``len``, ``keys/values/items``,
    ``cpu / cuda / to / detach / numpy``, ``in``."""

    def __init__(self, metainfo: Optional[dict] = None, **fields):
        object.__setattr__(self, "_f", dict(fields))
        object.__setattr__(self, "_meta", dict(metainfo or {}))

    def __getattr__(self, k):
        try:
            return self._f[k]
        except KeyError:
            if k in self._meta:
                return self._meta[k]
            raise AttributeError(f"InstanceData has no field {k!r}") from None

    def __setattr__(self, k, v):
        if self._f and hasattr(v, "__len__") and len(v) != len(self):
            raise AssertionError(f"the length of values {len(v)} is not consistent with the length of this InstanceData {len(self)}")
        self._f[k] = v

    def __delattr__(self, k):
        del self._f[k]

    def __getitem__(self, item):
        if isinstance(item, str):
            return self._f[item]
        if isinstance(item, int):
            if item >= len(self) or item < -len(self):
                raise IndexError(f"index {item} out of range")
            item = slice(item, None, len(self)) if item >= 0 else slice(item + len(self), None, len(self))
        return InstanceData(self._meta, **{k: (v[item] if not isinstance(v, list) else _index_list(v, item)) for k, v in self._f.items()})

    def __contains__(self, k):
        return k in self._f or k in self._meta

    def __len__(self):
        return len(next(iter(self._f.values()))) if self._f else 0

    def __repr__(self):
        return "<InstanceData(" + ", ".join(f"{k}: {tuple(v.shape) if hasattr(v, 'shape') else type(v).__name__}" for k, v in self._f.items()) + ")>"

    def keys(self):
        return list(self._f.keys())

    def values(self):
        return list(self._f.values())

    def items(self):
        return list(self._f.items())

    def get(self, k, default=None):
        return self._f.get(k, default)

    def _map(self, fn):
        return InstanceData(self._meta, **{k: (fn(v) if isinstance(v, torch.Tensor) else v) for k, v in self._f.items()})

    def cpu(self):
        return self._map(lambda t: t.cpu())

    def cuda(self):
        return self._map(lambda t: t.cuda())

    def to(self, *a, **kw):
        return self._map(lambda t: t.to(*a, **kw))

    def detach(self):
        return self._map(lambda t: t.detach())

    def numpy(self):
        return InstanceData(self._meta, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in self._f.items()})


def _index_list(v: list, item):
    if isinstance(item, slice):
        return v[item]
    idx = item.tolist() if hasattr(item, "tolist") else list(item)
    if idx and isinstance(idx[0], bool):
        return [x for x, m in zip(v, idx) if m]
    return [v[i] for i in idx]


class DetDataSample:
    """Data sample as ``PackDetInputs`` builds it: ``metainfo`` (also readable as attributes: ``sample.texts``,
    ``sample.ori_shape``, ...) + ``pred_instances`` / ``gt_instances``."""

    def __init__(self, metainfo: Optional[dict] = None, **kw):
        object.__setattr__(self, "_meta", dict(metainfo or {}))
        object.__setattr__(self, "_data", {})
        self._meta.update(kw)

    @property
    def metainfo(self) -> dict:
        return self._meta

    def set_metainfo(self, metainfo: dict) -> None:
        self._meta.update(metainfo)

    def metainfo_keys(self):
        return list(self._meta.keys())

    def keys(self):
        return list(self._data.keys())

    def __contains__(self, k):
        return k in self._data or k in self._meta

    def get(self, k, default=None):
        return self._data.get(k, self._meta.get(k, default))

    def __getattr__(self, k):
        if k in self._data:
            return self._data[k]
        if k in self._meta:
            return self._meta[k]
        if k in ("pred_instances", "gt_instances", "ignored_instances"):
            return None
        raise AttributeError(f"DetDataSample has no attribute {k!r}")

    def __setattr__(self, k, v):
        self._data[k] = v

    def _map(self, name, *a, **kw):
        out = DetDataSample(self._meta)
        for k, v in self._data.items():
            out._data[k] = getattr(v, name)(*a, **kw) if hasattr(v, name) else v
        return out

    def cpu(self):
        return self._map("cpu")

    def cuda(self):
        return self._map("cuda")

    def to(self, *a, **kw):
        return self._map("to", *a, **kw)

    def numpy(self):
        return self._map("numpy")


def _meta_of(sample) -> dict:
    if sample is None:
        return {}
    if isinstance(sample, dict):
        return sample.get("metainfo", sample)
    return getattr(sample, "metainfo", {}) or {}


def _flat_texts(texts) -> Tuple[str, ...]:
    """``[[a], [b]]`` (reparameterize, infer_wedetect.py:165-167), ``[a, b]`` (what LoadText leaves in a data sample,
    mm_transforms.py:126-133) -> ``(a, b)``: one caption per class, the first of each list."""
    return tuple((t[0] if isinstance(t, (list, tuple)) else t) for t in texts)


class MultiModalYOLOBackbone:
    """``model.backbone`` of the reference detector (mm_backbone.py:594-656) as far as inference callers touch it:
    ``forward_text(texts)`` and ``with_text_model``.  The image side is the fused ``ImageTower``; it has no
    per-module forward."""

    def __init__(self, image_model, text_model=None, frozen_stages: int = -1, with_text_model: bool = True, **kw):
        self.image_model = MODELS.build(image_model) if isinstance(image_model, dict) else image_model
        self.text_model = MODELS.build(text_model) if isinstance(text_model, dict) else text_model
        self.with_text_model = self.text_model is not None and with_text_model

    def forward_text(self, text: List[List[str]]) -> torch.Tensor:
        assert self.with_text_model, "forward_text() requires a text model"          # mm_backbone.py:648
        return self.text_model(text)

    def forward_image(self, image):
        raise NotImplementedError("the image tower is one fused launch sequence (engine.ImageTower): use "
                                  "YOLOWorldDetector.predict / test_step")


class YOLOWorldDetector(_DeviceModule):
    """Drop-in for the configured ``YOLOWorldDetector`` (mm_neck=False, use_bn_head=True).

    Two ways to construct it.  (1) The reference's keywords, as ``MODELS.build(cfg.model)`` passes them
    (yolo_world.py:15-24 + the mmdet single-stage detector's ``data_preprocessor, backbone, neck, bbox_head, train_cfg,
    test_cfg``): the dicts are validated by ``config.check_model_cfg`` — unknown registry names raise ``KeyError``,
    options outside the implemented path ``NotImplementedError`` — and the text tower named by
    ``backbone.text_model`` is wired in, so ``reparameterize(texts)`` works as at yolo_world.py:58-61.
    (2) ``YOLOWorldDetector("base", test_cfg=..., text_encoder=...)`` for callers without a config.

    ``test_cfg`` keeps the reference's keys (config/wedetect_base.py:18-25).  Inputs may have any H x W that is a
    multiple of 32 (one ImageTower is kept per shape); ``img_scale`` = (H, W) is only the default shape."""

    def __init__(self, model_size: Optional[str] = None, img_scale=None, test_cfg=None,
                 text_encoder: Optional[Callable] = None, max_classes: Optional[int] = None, precision: Optional[str] = None,
                 *, mm_neck: bool = False, num_train_classes: int = 80, num_test_classes: int = 80,
                 data_preprocessor=None, backbone=None, neck=None, bbox_head=None, train_cfg=None, init_cfg=None,
                 tokenizer=None):
        super().__init__()
        self.mm_neck, self.num_train_classes, self.num_test_classes = mm_neck, num_train_classes, num_test_classes
        self.backbone: Optional[MultiModalYOLOBackbone] = None
        if backbone is not None:
            from .config import check_model_cfg
            model_cfg = dict(type="YOLOWorldDetector", mm_neck=mm_neck, num_train_classes=num_train_classes,
                             num_test_classes=num_test_classes, data_preprocessor=data_preprocessor, backbone=backbone,
                             neck=neck, bbox_head=bbox_head, train_cfg=train_cfg, test_cfg=test_cfg)
            size = check_model_cfg(model_cfg)
            if model_size is not None and model_size != size:
                raise ValueError(f"model_size={model_size!r} contradicts the config ({size!r})")
            model_size = size
            self.backbone = MultiModalYOLOBackbone(backbone["image_model"], backbone.get("text_model"))
            if self.backbone.text_model is not None:
                if tokenizer is not None:
                    self.backbone.text_model.tokenizer = tokenizer
                if precision is not None:
                    self.backbone.text_model.precision = precision
            if max_classes is None:
                max_classes = max(int(num_test_classes), 1)
        elif model_size is None:
            raise TypeError("YOLOWorldDetector needs either the config keywords (backbone=..., neck=..., bbox_head=...) or a model_size")
        cfg = dict(multi_label=True, nms_pre=30000, score_thr=0.001, nms=dict(type="nms", iou_threshold=0.7),
                   max_per_img=300)
        cfg.update(test_cfg or {})
        if not cfg["multi_label"]:
            raise NotImplementedError("only multi_label=True (every shipped config) is implemented")
        nms = dict(cfg["nms"])
        if nms.get("type", "nms") != "nms":
            raise NotImplementedError(f"test_cfg.nms.type={nms.get('type')!r}: only the plain greedy 'nms' is implemented")
        # mmcv.ops.batched_nms options (mmcv/ops/nms.py 2.1.0).  max_num: ``keep[:max_num]`` after NMS — followed by mmdet's
        # ``results[:max_per_img]``, i.e. min(max_num, max_per_img) rows.  score_threshold: ``scores > score_threshold`` before
        # NMS — the candidates already passed ``scores > score_thr`` and the nms_pre cut keeps the highest, so the two filters are
        # one with the larger threshold.  class_agnostic: NMS on the un-offset boxes across classes — not built.
        if nms.get("class_agnostic", False):
            raise NotImplementedError("test_cfg.nms.class_agnostic=True (mmcv.ops.batched_nms on un-offset boxes across classes) "
                                      "is not implemented; every shipped config uses class-aware NMS")
        if set(nms) - {"type", "iou_threshold", "split_thr", "class_agnostic", "max_num", "score_threshold"}:
            raise NotImplementedError("test_cfg.nms options "
                                      f"{sorted(set(nms) - {'type', 'iou_threshold', 'split_thr', 'class_agnostic', 'max_num', 'score_threshold'})} are not implemented")
        max_num = int(nms.get("max_num", -1))
        if max_num > 0:
            cfg["max_per_img"] = min(int(cfg["max_per_img"]), max_num)
        if float(nms.get("score_threshold", 0.0)) > float(cfg["score_thr"]):
            # mmcv decides between its one-call and its per-class branch on the candidate count BEFORE this filter
            # (boxes.shape[0] < split_thr, mmcv/ops/nms.py batched_nms), the merged filter decides on the count after it: the two
            # can only differ when nms_pre candidates can reach split_thr — then the option is refused rather than emulated
            # with last-bit IoU differences.  No shipped config sets score_threshold.
            if int(cfg["nms_pre"]) >= int(nms.get("split_thr", 10000)):
                raise NotImplementedError("test_cfg.nms.score_threshold above score_thr with nms_pre >= split_thr: mmcv picks its NMS "
                                          "branch on the pre-filter candidate count, which the merged filter cannot reproduce; raise "
                                          "score_thr instead, or lower nms_pre below split_thr")
            cfg["score_thr"] = float(nms["score_threshold"])
        self.test_cfg = cfg
        self.model_size = model_size
        self.img_scale = tuple(img_scale) if img_scale is not None else _IMG_SIZE[model_size]
        self._h = _TowerHolder(model_size, 0, max_classes if max_classes is not None else 1203, int(cfg["max_per_img"]),
                               precision, nms_pre=int(cfg["nms_pre"]))
        self.text_encoder = text_encoder
        self.texts = None
        self.text_feats: Optional[torch.Tensor] = None
        self._banks: Dict[Tuple[str, ...], torch.Tensor] = {}

    # -- weights --------------------------------------------------------------------------
    def _load_state(self, state_dict, strict: bool):
        msg = self._h.load(state_dict, strict)
        tm = self.backbone.text_model if self.backbone is not None else None
        if tm is not None:
            sub = {k: v for k, v in state_dict.items() if k.startswith("backbone.text_model.")}
            if sub:
                tm.load_state_dict(sub)
            elif strict:
                raise RuntimeError("Error(s) in loading state_dict: no backbone.text_model.* tensors in the checkpoint")
        self._banks.clear()
        self.text_feats = self.texts = None
        return msg

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = super().state_dict(destination=destination, prefix=prefix)
        tm = self.backbone.text_model if self.backbone is not None else None
        if tm is not None:
            for k, v in tm.state_dict().items():
                out[prefix + "backbone.text_model." + k] = v
        return out

    def _on_device(self, device) -> None:
        tm = self.backbone.text_model if self.backbone is not None else None
        if tm is not None:
            tm.cuda(device)

    # -- text side ------------------------------------------------------------------------
    def set_text_embeddings(self, feats: torch.Tensor, texts: Optional[List[List[str]]] = None) -> None:
        """Class bank [K, 768] (any norm: it is L2-normalised on device like
        BNContrastiveHead does, yolo_world_head.py:101)."""
        if feats.dim() == 3:
            if feats.shape[0] != 1:
                raise ValueError("a [B, K, 768] bank must have B == 1 (one bank for the whole batch)")
            feats = feats[0]
        if feats.dim() != 2 or feats.shape[1] != EMBED_DIM:
            raise ValueError("text embeddings must be [K, 768]")
        self.text_feats = feats.detach().to(torch.float32)
        self.texts = texts
        if texts is not None:
            self._banks[_flat_texts(texts)] = self.text_feats

    def _encode(self, flat: Tuple[str, ...]) -> torch.Tensor:
        if self.backbone is not None and self.backbone.with_text_model:
            return self.backbone.forward_text([list(flat)])[0]              # [1, K, D] -> [K, D]
        if self.text_encoder is not None:
            return self.text_encoder(list(flat))
        raise NotImplementedError("no text tower: build the detector from a config with backbone.text_model, pass "
                                  "text_encoder=..., or call set_text_embeddings(bank)")

    def reparameterize(self, texts: List[List[str]]) -> None:
        """yolo_world.py:58-61: ``self.texts = texts; self.text_feats = self.backbone.forward_text(texts)``."""
        flat = _flat_texts(texts)
        self.set_text_embeddings(self._encode(flat), texts)

    def _bank_for(self, sample) -> torch.Tensor:
        """extract_feat (yolo_world.py:82-113): data samples that carry ``texts`` are scored against those texts; the
        reference re-runs the text tower for every image, here the bank of each distinct class list is built once."""
        texts = _meta_of(sample).get("texts") if sample is not None else None
        if texts is None:
            if self.text_feats is None:
                raise RuntimeError("no class bank: call reparameterize(texts) or set_text_embeddings(bank) first, or "
                                   "pack `texts` into the data samples")
            return self.text_feats
        flat = _flat_texts(texts)
        bank = self._banks.get(flat)
        if bank is None:
            if len(self._banks) > 64:
                self._banks.clear()
            bank = self._banks[flat] = self._encode(flat).detach().to(torch.float32)
        return bank

    # -- image side -----------------------------------------------------------------------
    @torch.no_grad()
    def test_step(self, data: dict):
        inputs, samples = data["inputs"], data.get("data_samples")
        if samples is None:
            samples = [None] * len(inputs)
        return self.predict(inputs, samples)

    def forward(self, inputs, data_samples=None, mode: str = "predict"):
        if mode != "predict":
            raise NotImplementedError(f"mode={mode!r}: only inference ('predict') is on this path")
        return self.predict(inputs, data_samples if data_samples is not None else [None] * len(inputs))

    @torch.no_grad()
    def predict(self, batch_inputs, batch_data_samples, rescale: bool = True):
        """batch_inputs: list / tensor of [3, H, W] images as the mmdet pipeline packs them
        (uint8 or float 0-255, BGR — DetDataPreprocessor does bgr_to_rgb and /255,
        wedetect_base.py:44-48), all of one shape with H, W multiples of 32 (the test pipeline letterboxes
        to ``img_scale``)."""
        dev = self._h.device
        if dev is None:
            raise RuntimeError("model is not on a HIP device: call .cuda()")
        xs = [x for x in batch_inputs]
        samples = list(batch_data_samples) if batch_data_samples is not None else [None] * len(xs)
        if len(samples) != len(xs):
            raise ValueError(f"{len(xs)} inputs but {len(samples)} data samples")
        banks = [self._bank_for(s) for s in samples]
        shapes = {tuple(t.shape) for t in xs}
        if len(shapes) != 1 or len(next(iter(shapes))) != 3 or next(iter(shapes))[0] != 3:
            raise ValueError(f"inputs must be [3, H, W] tensors of one shape after the test pipeline, got {sorted(shapes)}")
        _, hh, ww = next(iter(shapes))
        if hh % 32 or ww % 32:
            raise ValueError(f"input size {hh}x{ww} is not a multiple of 32 (letterbox to img_scale first)")
        out: List[Optional[DetDataSample]] = [None] * len(xs)
        groups: Dict[int, List[int]] = {}
        for i, bk in enumerate(banks):
            groups.setdefault(id(bk), []).append(i)
        for idxs in groups.values():
            from . import lib as L
            chw = torch.stack([xs[i].to(dev) for i in idxs])     # [b, 3, H, W] BGR, contiguous
            if chw.dtype not in (torch.uint8, torch.float32):
                chw = chw.to(torch.float32)
            x = torch.empty(len(idxs), hh, ww, 3, dtype=torch.uint8, device=dev)
            L.chw_to_hwc_u8(chw, x)                              # bgr_to_rgb + NHWC (data_preprocessor.py:35-36)
            metas = []
            for i in idxs:
                m = _meta_of(samples[i])
                ori = m.get("ori_shape", (hh, ww))
                sf = m.get("scale_factor", (1.0, 1.0))
                pad = m.get("pad_param", None)
                px, py = (0.0, 0.0) if pad is None else (float(pad[2]), float(pad[0]))
                sx, sy = (float(sf[0]), float(sf[1])) if rescale else (1.0, 1.0)
                if not rescale:
                    px = py = 0.0
                metas.append([px, py, 0.0, sx, sy, float(ori[1]), float(ori[0]), 1.0])
            tower = self._h.tower(len(idxs), hh, ww)
            meta = torch.tensor(metas, dtype=torch.float32, device=dev)
            bank = banks[idxs[0]].to(dev)
            run = lambda: self._h.detect(tower, x, bank, meta, normalize_text=True, score_thr=self.test_cfg["score_thr"],
                                         iou_thr=self.test_cfg["nms"]["iou_threshold"], with_embed=False,
                                         # mmdet _bbox_post_process -> mmcv.ops.batched_nms(bboxes, scores, labels, cfg.nms)
                                         nms="mmcv", nms_param=int(self.test_cfg["nms"].get("split_thr", 10000)))
            res = run()
            recal = (lambda: self._h.recalibrate(tower, x)) if self._h.auto_calibrate else None
            counts = tower.checked_counts(res, run, recal)
            self._h.precision = "fp32" if tower.overflowed else self._h._asked_precision
            for j, (i, n) in enumerate(zip(idxs, counts)):
                inst = InstanceData(bboxes=res["bboxes"][j, :n].clone(), scores=res["scores"][j, :n].clone(),
                                    labels=res["labels"][j, :n].to(torch.int64))
                s = samples[i]
                if s is None:
                    s = DetDataSample()
                elif isinstance(s, dict):
                    s = DetDataSample(metainfo=_meta_of(s))
                s.pred_instances = inst
                out[i] = s
        return out


MODELS.register_module(module=YOLOWorldDetector)
MODELS.register_module(module=MultiModalYOLOBackbone)
