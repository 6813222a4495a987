"""Operator surface of the hot path — the classes callers of the reference construct.

Mirrors (names, arguments, return structures, error behaviour):
  * ``SimpleYOLOWorldDetector``  generate_proposal.py:1052-1218 / extract_embedding.py:1088-1262
        model = SimpleYOLOWorldDetector('base', prompt_dim=768, num_prompts=256)
        model.load_state_dict(ckpt, strict=False); model.cuda(); model.eval()
        outputs = model([path_or_PIL, ...])  ->  list of dict(bboxes, embeddings, scores, labels, scales, bias)
  * ``YOLOWorldDetector``        wedetect/models/detectors/yolo_world.py:19-113
        model.reparameterize(texts) / model.test_step(dict(inputs=[...], data_samples=[...]))
        -> data samples with ``.pred_instances.{bboxes, scores, labels}`` (infer_wedetect.py:117-131)

All tensor math is done by libwedetect_hip.so through ``ImageTower``; this file only does
host-side bookkeeping (PIL letterbox, metadata, result containers).  There is no CPU
fallback: constructing a detector without a HIP device / the built library raises.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .arch import EMBED_DIM, all_params, get_arch

_IMG_SIZE = {"tiny": (640, 640), "base": (640, 640), "large": (1280, 1280), "nano": (128, 128)}


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
    """Shared plumbing: weights -> packed device tensors -> ImageTower per batch size."""

    def __init__(self, arch: str, num_prompts: int, img_size, max_classes: int, max_out: int,
                 precision: Optional[str] = None):
        self.arch = get_arch(arch)
        self.precision = precision
        self.num_prompts = num_prompts
        self.img_size = tuple(img_size)
        self.max_classes, self.max_out = max_classes, max_out
        self._sd: Optional[Dict[str, np.ndarray]] = None
        self._packed = None
        self._towers = {}
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
        return _IncompatibleKeys(missing, unexpected)

    def tower(self, batch: int):
        from .engine import ImageTower
        from .pack import pack
        if self._sd is None:
            raise RuntimeError("load_state_dict() must be called before inference")
        if self.device is None:
            raise RuntimeError("model is not on a HIP device: call .cuda() (there is no CPU execution path)")
        if self._packed is None:
            self._packed = pack(self._sd, self.arch, self.device)
        if batch not in self._towers:
            self._towers[batch] = ImageTower(self.arch, self._packed, batch, self.img_size[0], self.img_size[1],
                                             device=self.device, max_classes=self.max_classes, max_out=self.max_out,
                                             precision=self.precision)
        return self._towers[batch]


# ------------------------------------------------------------------------------------------
# WeDetect-Uni: proposals + embeddings
# ------------------------------------------------------------------------------------------
class SimpleYOLOWorldDetector:
    """Drop-in for generate_proposal.py:1052 / extract_embedding.py:1088."""

    def __init__(self, backbone_size, prompt_dim=768, num_prompts=512, num_proposals=300, img_size=None,
                 precision: Optional[str] = None):
        if prompt_dim != EMBED_DIM:
            raise ValueError("prompt_dim must be 768")
        self.backbone_size = backbone_size
        self.num_proposals = num_proposals
        self.img_size = tuple(img_size) if img_size is not None else _IMG_SIZE[backbone_size]
        self._h = _TowerHolder(backbone_size, num_prompts, self.img_size, max(num_prompts, 1), num_proposals, precision)
        self._lb = None
        self.training = False

    # -- nn.Module-like surface used by the reference scripts
    def load_state_dict(self, state_dict, strict: bool = True):
        return self._h.load(state_dict, strict)

    def cuda(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: wedetect_amd has no CPU execution path")
        self._h.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        return self

    def eval(self):
        self.training = False
        return self

    def __call__(self, image_paths, rescale=True):
        return self.forward(image_paths, rescale)

    @torch.no_grad()
    def forward(self, image_paths: Sequence[Union[str, object]], rescale=True) -> List[Dict[str, torch.Tensor]]:
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
            w, h = img.size
            sc = ratio if rescale else 1.0
            metas.append([dw, dh, 0.0, sc, sc, float(w), float(h), 0.0])
        b = len(imgs)
        tower = self._h.tower(b)
        meta = torch.tensor(metas, dtype=torch.float32, device=self._h.device)
        res = tower.detect(x, tower.P["prompts"], meta, normalize_text=False, score_thr=0.0, with_embed=True)
        counts = res["count"].tolist()                      # one D2H sync per batch
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


# ------------------------------------------------------------------------------------------
# WeDetect: open-vocabulary detection against a text bank
# ------------------------------------------------------------------------------------------
class InstanceData:
    """Minimal stand-in for mmengine.structures.InstanceData as infer_wedetect.py uses it:
    attribute and item access, boolean-mask / index ``__getitem__``, ``cpu()``, ``numpy()``."""

    def __init__(self, **fields):
        object.__setattr__(self, "_f", dict(fields))

    def __getattr__(self, k):
        try:
            return self._f[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._f[k] = v

    def __getitem__(self, item):
        if isinstance(item, str):
            return self._f[item]
        return InstanceData(**{k: v[item] for k, v in self._f.items()})

    def __contains__(self, k):
        return k in self._f

    def __len__(self):
        return len(next(iter(self._f.values()))) if self._f else 0

    def keys(self):
        return self._f.keys()

    def cpu(self):
        return InstanceData(**{k: v.cpu() for k, v in self._f.items()})

    def numpy(self):
        return InstanceData(**{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in self._f.items()})


class YOLOWorldDetector:
    """Drop-in for the configured ``YOLOWorldDetector`` (mm_neck=False, use_bn_head=True).

    ``test_cfg`` keeps the reference's keys (config/wedetect_base.py:18-25).  The class bank comes
    from a text tower: pass ``text_encoder`` (callable List[str] -> [K, 768] tensor, e.g.
    ``wedetect_amd.text.XLMRobertaLanguageBackbone(...).encode_classes``) or set the bank directly
    with ``set_text_embeddings``; ``reparameterize`` without either raises."""

    def __init__(self, model_size="base", img_scale=None, test_cfg=None, text_encoder: Optional[Callable] = None,
                 max_classes: int = 1203, precision: Optional[str] = None):
        cfg = dict(multi_label=True, nms_pre=30000, score_thr=0.001, nms=dict(type="nms", iou_threshold=0.7),
                   max_per_img=300)
        cfg.update(test_cfg or {})
        if not cfg["multi_label"]:
            raise NotImplementedError("only multi_label=True (every shipped config) is implemented")
        self.test_cfg = cfg
        self.model_size = model_size
        self.img_scale = tuple(img_scale) if img_scale is not None else _IMG_SIZE[model_size]
        self._h = _TowerHolder(model_size, 0, self.img_scale, max_classes, cfg["max_per_img"], precision)
        self.text_encoder = text_encoder
        self.texts = None
        self.text_feats: Optional[torch.Tensor] = None
        self.training = False

    def load_state_dict(self, state_dict, strict: bool = True):
        if "state_dict" in state_dict and isinstance(state_dict["state_dict"], dict):
            state_dict = state_dict["state_dict"]           # mmengine checkpoint wrapper
        return self._h.load(state_dict, strict)

    def cuda(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: wedetect_amd has no CPU execution path")
        self._h.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        return self

    to = lambda self, device: self.cuda()

    def eval(self):
        self.training = False
        return self

    # -- text side ------------------------------------------------------------------------
    def set_text_embeddings(self, feats: torch.Tensor, texts: Optional[List[List[str]]] = None) -> None:
        """Class bank [K, 768] (any norm: it is L2-normalised on device like
        BNContrastiveHead does, yolo_world_head.py:101)."""
        if feats.dim() == 3:
            feats = feats[0]
        if feats.dim() != 2 or feats.shape[1] != EMBED_DIM:
            raise ValueError("text embeddings must be [K, 768]")
        self.text_feats = feats.detach().to(torch.float32)
        self.texts = texts

    def reparameterize(self, texts: List[List[str]]) -> None:
        """yolo_world.py:58-61.  Needs the text tower, which is supplied by the caller."""
        if self.text_encoder is None:
            raise NotImplementedError("reparameterize(texts) needs text_encoder=... (wedetect_amd.text.XLMRobertaLanguageBackbone"
                                      " with a tokenizer) or set_text_embeddings(bank)")
        self.set_text_embeddings(self.text_encoder([t[0] if isinstance(t, (list, tuple)) else t for t in texts]), texts)

    # -- image side -----------------------------------------------------------------------
    @torch.no_grad()
    def test_step(self, data: dict):
        inputs, samples = data["inputs"], data.get("data_samples")
        if samples is None:
            samples = [None] * len(inputs)
        return self.predict(inputs, samples)

    @torch.no_grad()
    def predict(self, batch_inputs, batch_data_samples, rescale: bool = True):
        """batch_inputs: list / tensor of [3, H, W] images as the mmdet pipeline packs them
        (uint8 or float 0-255, BGR — DetDataPreprocessor does bgr_to_rgb and /255,
        wedetect_base.py:44-48).  Every image must already be letterboxed to ``img_scale``."""
        if self.text_feats is None:
            raise RuntimeError("no class bank: call reparameterize(texts) or set_text_embeddings(bank) first")
        dev = self._h.device
        if dev is None:
            raise RuntimeError("model is not on a HIP device: call .cuda()")
        xs = [x for x in batch_inputs]
        b = len(xs)
        x = torch.stack([t.to(dev) for t in xs])
        if tuple(x.shape[1:]) != (3, self.img_scale[0], self.img_scale[1]):
            raise ValueError(f"inputs must be [3,{self.img_scale[0]},{self.img_scale[1]}] after the test pipeline")
        x = x.flip(1).permute(0, 2, 3, 1)                    # BGR CHW -> RGB HWC
        x = (x if x.dtype == torch.uint8 else x.round().clamp(0, 255).to(torch.uint8)).contiguous()
        metas = []
        for s in batch_data_samples:
            m = _meta_of(s)
            ori = m.get("ori_shape", self.img_scale)
            sf = m.get("scale_factor", (1.0, 1.0))
            pad = m.get("pad_param", None)
            px, py = (0.0, 0.0) if pad is None else (float(pad[2]), float(pad[0]))
            sx, sy = (float(sf[0]), float(sf[1])) if rescale else (1.0, 1.0)
            if not rescale:
                px = py = 0.0
            metas.append([px, py, 0.0, sx, sy, float(ori[1]), float(ori[0]), 1.0])
        tower = self._h.tower(b)
        meta = torch.tensor(metas, dtype=torch.float32, device=dev)
        bank = self.text_feats.to(dev)
        res = tower.detect(x, bank, meta, normalize_text=True, score_thr=self.test_cfg["score_thr"],
                           iou_thr=self.test_cfg["nms"]["iou_threshold"], with_embed=False)
        counts = res["count"].tolist()
        out = []
        for i, (n, s) in enumerate(zip(counts, batch_data_samples)):
            inst = InstanceData(bboxes=res["bboxes"][i, :n].clone(), scores=res["scores"][i, :n].clone(),
                                labels=res["labels"][i, :n].to(torch.int64))
            if s is None:
                s = DetDataSample()
            s.pred_instances = inst
            out.append(s)
        return out


class DetDataSample:
    """Bare data sample: ``metainfo`` dict + ``pred_instances``."""

    def __init__(self, metainfo: Optional[dict] = None, **kw):
        self.metainfo = dict(metainfo or {})
        self.metainfo.update(kw)
        self.pred_instances = None

    @property
    def texts(self):
        return self.metainfo.get("texts")


def _meta_of(sample) -> dict:
    if sample is None:
        return {}
    if isinstance(sample, dict):
        return sample.get("metainfo", sample)
    return getattr(sample, "metainfo", {}) or {}
