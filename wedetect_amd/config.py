"""Config-driven construction: the reference's ``model = dict(type="YOLOWorldDetector", ...)`` -> the device detector.

``infer_wedetect.py`` builds its model with mmdet's ``init_detector(config, checkpoint)``, i.e.
``MODELS.build(cfg.model)`` over the dict in ``config/wedetect_{tiny,base,large}.py:39-107``.  ``build_detector``
takes that same dict (plain Python: the config files only need ``exec``, or mmengine's ``Config`` where it exists):
every registry ``type`` name the shipped configs use resolves here, every option this path implements is honoured,
and every option it does NOT implement is refused loudly instead of being ignored (``mm_neck=True``,
``use_bn_head=False``, other strides / offsets / means) — a config that builds here runs the same network as the
reference would.  Training-only entries (losses, assigner, ``train_cfg``) must carry known type names and are
otherwise unused, as in ``model.eval()`` inference.

The test pipeline (``test_pipeline`` in the same files) is host work around the model: ``pipeline_plan`` maps its
transform names onto what this package does instead (device letterbox, no-ops)."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

from .arch import CLS_MID, EMBED_DIM, REG_MAX, REG_MID, STRIDES, get_arch
from .registry import MODELS

# registry type names of config/wedetect_*.py -> what they are here
MODEL_TYPES = {
    "YOLOWorldDetector": "wedetect_amd.detector.YOLOWorldDetector",
    "YOLOWDetDataPreprocessor": "stem kernel: /255 and BGR->RGB (wd_stem_patchify; detector.predict flips the channels)",
    "MultiModalYOLOBackbone": "image tower (engine.ImageTower) + text tower (text.XLMRobertaLanguageBackbone)",
    "ConvNextVisionBackbone": "engine.ImageTower.backbone",
    "XLMRobertaLanguageBackbone": "wedetect_amd.text.XLMRobertaLanguageBackbone",
    "CSPRepBiFPANNeck": "engine.ImageTower.neck",
    "YOLOWorldHead": "engine.ImageTower.head + similarity + postprocess",
    "YOLOWorldHeadModule": "engine.ImageTower.head",
    "MlvlPointGenerator": "wd_dfl_decode (priors (i + 0.5) * stride)",
    "WeDetectDistancePointBBoxCoder": "wd_dfl_decode (ltrb * stride -> xyxy)",
}
TRAINING_ONLY_TYPES = ("CrossEntropyLoss", "mmyoloIoULoss", "DistributionFocalLoss", "BatchTaskAlignedAssigner")
PIPELINE_TYPES = {
    "LoadImageFromFile": "host: read the image (PIL / cv2), uint8 HWC",
    "WeDetectKeepRatioResize": "pipeline.WeDetectKeepRatioResize: the reference's geometry (transforms.py:94-123), pixels resampled on "
                               "the DEVICE by wd_cv_resize_paste_u8 (OpenCV's INTER_AREA / INTER_LINEAR arithmetic, fused with the pad)",
    "WeDetectLetterResize": "pipeline.WeDetectLetterResize: pad_param / scale_factor bookkeeping of transforms.py:180-272; the 114 pad is "
                            "written by the same device kernel",
    "LoadAnnotations": "no-op at inference",
    "LoadText": "class names -> YOLOWorldDetector.reparameterize(texts)",
    "PackDetInputs": "detector.DetDataSample metainfo (ori_shape, scale_factor, pad_param)",
}


def _typed(cfg: dict, where: str, expected: str) -> dict:
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise KeyError(f"{where}: expected a config dict with a 'type' key")
    if cfg["type"] != expected:
        known = expected in MODEL_TYPES
        raise KeyError(f"{where}: type {cfg['type']!r} is not in the wedetect_amd registry"
                       + (f" (this slot takes {expected!r})" if known else ""))
    return cfg


def _require(cond: bool, what: str) -> None:
    if not cond:
        raise NotImplementedError(f"config option outside the implemented path: {what}")


def model_size_of(model_cfg: dict) -> str:
    """The single size name ('tiny' | 'base' | 'large') a consistent config carries in four places."""
    bb = _typed(model_cfg["backbone"], "model.backbone", "MultiModalYOLOBackbone")
    im = _typed(bb["image_model"], "model.backbone.image_model", "ConvNextVisionBackbone")
    size = im["model_name"]
    arch = get_arch(size)
    neck = _typed(model_cfg["neck"], "model.neck", "CSPRepBiFPANNeck")
    head = _typed(model_cfg["bbox_head"], "model.bbox_head", "YOLOWorldHead")
    hm = _typed(head["head_module"], "model.bbox_head.head_module", "YOLOWorldHeadModule")
    names = {"image_model.model_name": size, "neck.model_size": neck.get("model_size", size),
             "head_module.model_size": hm.get("model_size", size)}
    if "text_model" in bb and bb["text_model"] is not None:
        tm = _typed(bb["text_model"], "model.backbone.text_model", "XLMRobertaLanguageBackbone")
        names["text_model.model_size"] = tm.get("model_size", size)
    if len(set(names.values())) != 1:
        raise ValueError(f"inconsistent model sizes in the config: {names}")
    # CSPRepBiFPANNeck: repeats from model_size, widths from scale_factor (constructor default 0.75,
    # yolo_world_pafpn.py:992 — what the tiny config relies on); both must describe the same size
    sf = neck.get("scale_factor", 0.75)
    _require(float(sf) == float(arch.neck_scale), f"neck.scale_factor {sf} for size {size!r} (expected {arch.neck_scale})")
    return size


def check_model_cfg(model_cfg: dict) -> str:
    """Validates every entry of the reference's ``model`` dict against what the device path implements; returns the
    model size.  KeyError: unknown registry name (what mmengine raises); NotImplementedError: a known option outside
    the implemented path; ValueError: inconsistent sizes."""
    _typed(model_cfg, "model", "YOLOWorldDetector")
    _require(not model_cfg.get("mm_neck", False), "mm_neck=True (text-guided neck; the bricks exist in wedetect_amd.bricks, "
             "the assembled YOLOWorldPAFPN does not)")
    size = model_size_of(model_cfg)
    arch = get_arch(size)
    dp = model_cfg.get("data_preprocessor")
    if dp is not None:
        _typed(dp, "model.data_preprocessor", "YOLOWDetDataPreprocessor")
        _require([float(v) for v in dp.get("mean", [0.0] * 3)] == [0.0] * 3, f"data_preprocessor.mean {dp.get('mean')}")
        _require([float(v) for v in dp.get("std", [255.0] * 3)] == [255.0] * 3, f"data_preprocessor.std {dp.get('std')}")
        _require(bool(dp.get("bgr_to_rgb", True)), "data_preprocessor.bgr_to_rgb=False")
    head = model_cfg["bbox_head"]
    hm = head["head_module"]
    _require(bool(hm.get("use_bn_head", False)), "head_module.use_bn_head=False (ContrastiveHead without BatchNorm)")
    _require(int(hm.get("embed_dims", EMBED_DIM)) == EMBED_DIM, f"head_module.embed_dims {hm.get('embed_dims')}")
    # the module takes its input widths from model_size and only its branch widths from these two
    # (yolo_world_head.py:178-190): cls = max(in_channels[0], num_classes), reg = max(16, in_channels[0] // 4, 4 * reg_max)
    in0, ncls = int(hm.get("in_channels", [256])[0]), int(hm.get("num_classes", 80))
    _require(max(in0, ncls) == CLS_MID and max(16, in0 // 4, 4 * REG_MAX) == REG_MID,
             f"head_module.in_channels[0] = {in0}, num_classes = {ncls} (branch widths {max(in0, ncls)} / "
             f"{max(16, in0 // 4, 4 * REG_MAX)}; built: {CLS_MID} / {REG_MID})")
    pg = head.get("prior_generator")
    if pg is not None:
        _typed(pg, "model.bbox_head.prior_generator", "MlvlPointGenerator")
        _require(float(pg.get("offset", 0.5)) == 0.5 and tuple(pg.get("strides", STRIDES)) == STRIDES,
                 f"prior_generator {pg}")
    bc = head.get("bbox_coder")
    if bc is not None:
        _typed(bc, "model.bbox_head.bbox_coder", "WeDetectDistancePointBBoxCoder")
    for key in ("loss_cls", "loss_bbox", "loss_dfl"):
        if head.get(key) is not None and head[key].get("type") not in TRAINING_ONLY_TYPES:
            raise KeyError(f"model.bbox_head.{key}: type {head[key].get('type')!r} is not in the wedetect_amd registry")
    asg = (model_cfg.get("train_cfg") or {}).get("assigner")
    if asg is not None and asg.get("type") not in TRAINING_ONLY_TYPES:
        raise KeyError(f"model.train_cfg.assigner: type {asg.get('type')!r} is not in the wedetect_amd registry")
    return size


def build_detector(model_cfg: dict, img_scale: Optional[Sequence[int]] = None, text_encoder: Optional[Callable] = None,
                   precision: Optional[str] = None, tokenizer=None):
    """``MODELS.build(cfg.model)`` for this package: the reference's model dict -> ``YOLOWorldDetector`` (not yet on a
    device, no weights: ``load_state_dict`` / ``.cuda()`` / ``reparameterize`` follow as in infer_wedetect.py:102-116),
    with the text tower the config names wired in.  ``img_scale``: the config's ``img_scale``, written (w, h) as
    mmdet does (config/wedetect_base.py:109, transforms.py:190 reverses it); kept as the default (H, W) input shape."""
    size = check_model_cfg(model_cfg)
    hw = None
    if img_scale is not None:
        if len(img_scale) != 2:
            raise TypeError("img_scale must be a (w, h) pair")
        hw = (int(img_scale[1]), int(img_scale[0]))
    det = MODELS.build(dict(model_cfg), img_scale=hw, text_encoder=text_encoder, precision=precision, tokenizer=tokenizer)
    assert det.model_size == size
    return det


def pipeline_plan(test_pipeline: Sequence[dict]) -> Dict[str, str]:
    """Transform type -> what stands in for it here; unknown transform names raise KeyError (as the registry would)."""
    plan = {}
    for step in test_pipeline:
        t = step.get("type")
        if t not in PIPELINE_TYPES:
            raise KeyError(f"test_pipeline: type {t!r} is not in the wedetect_amd registry")
        if t == "WeDetectLetterResize":
            pv = step.get("pad_val", dict(img=114))
            _require(int(pv.get("img", 114) if isinstance(pv, dict) else pv) == 114, f"pad_val {pv}")
            _require(not step.get("allow_scale_up", False), "WeDetectLetterResize.allow_scale_up=True")
        plan[t] = PIPELINE_TYPES[t]
    return plan


# ------------------------------------------------------------------------------------------------------
# registry entries for the component names of the configs.  The image side is ONE fused launch sequence
# (engine.ImageTower), so these are specification objects: they take the reference constructors' keywords,
# keep them, and refuse at construction what the tower does not implement — ``MODELS.build`` of any
# sub-dict of a shipped config resolves, and a stock mmengine registry (register_with_mmengine) finds the
# same names.  YOLOWorldDetector consumes the dicts directly (check_model_cfg above).
# ------------------------------------------------------------------------------------------------------
class _Spec:
    TYPE = ""

    def __init__(self, **kw):
        self.cfg = dict(kw, type=self.TYPE)

    def __repr__(self):
        return f"{self.TYPE}({', '.join(f'{k}={v!r}' for k, v in self.cfg.items() if k != 'type')})"

    def __call__(self, *a, **kw):
        raise NotImplementedError(f"{self.TYPE} is part of the fused image tower (wedetect_amd.engine.ImageTower) and has no "
                                  "stand-alone forward; run YOLOWorldDetector.predict / test_step")


def _spec(name: str, check=None, doc: str = ""):
    def __init__(self, **kw):
        _Spec.__init__(self, **kw)
        if check is not None:
            check(self.cfg)
    cls = type(name, (_Spec,), dict(TYPE=name, __init__=__init__, __doc__=doc or MODEL_TYPES.get(name, "")))
    MODELS.register_module(module=cls)
    return cls


def _check_image_model(c):
    get_arch(c["model_name"])


def _check_neck(c):
    size = c.get("model_size")
    if size is not None:
        _require(float(c.get("scale_factor", 0.75)) == float(get_arch(size).neck_scale),
                 f"neck.scale_factor {c.get('scale_factor', 0.75)} for size {size!r}")


def _check_head_module(c):
    _require(bool(c.get("use_bn_head", False)), "head_module.use_bn_head=False (ContrastiveHead without BatchNorm)")
    _require(int(c.get("embed_dims", EMBED_DIM)) == EMBED_DIM, f"head_module.embed_dims {c.get('embed_dims')}")


def _check_preproc(c):
    _require([float(v) for v in c.get("mean", [0.0] * 3)] == [0.0] * 3, f"data_preprocessor.mean {c.get('mean')}")
    _require([float(v) for v in c.get("std", [255.0] * 3)] == [255.0] * 3, f"data_preprocessor.std {c.get('std')}")
    _require(bool(c.get("bgr_to_rgb", True)), "data_preprocessor.bgr_to_rgb=False")


def _check_priors(c):
    _require(float(c.get("offset", 0.5)) == 0.5 and tuple(c.get("strides", STRIDES)) == STRIDES, f"prior_generator {c}")


ConvNextVisionBackbone = _spec("ConvNextVisionBackbone", _check_image_model)
CSPRepBiFPANNeck = _spec("CSPRepBiFPANNeck", _check_neck)
YOLOWorldHeadModule = _spec("YOLOWorldHeadModule", _check_head_module)
YOLOWorldHead = _spec("YOLOWorldHead")
YOLOWDetDataPreprocessor = _spec("YOLOWDetDataPreprocessor", _check_preproc)
MlvlPointGenerator = _spec("MlvlPointGenerator", _check_priors)
WeDetectDistancePointBBoxCoder = _spec("WeDetectDistancePointBBoxCoder")
for _n in TRAINING_ONLY_TYPES:
    _spec(_n, doc="training-only entry of the configs: constructed and kept, never evaluated on the inference path")


def _register_text_backbone():
    from .text import XLMRobertaLanguageBackbone
    MODELS.register_module(module=XLMRobertaLanguageBackbone)


_register_text_backbone()
from . import detector as _detector  # noqa: E402,F401  (registers YOLOWorldDetector / MultiModalYOLOBackbone)
