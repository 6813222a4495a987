"""``init_detector`` / ``inference_detector``: what infer_wedetect.py:102-131,150-160 imports from mmdet.apis and
defines itself, on the device path.

    cfg = Config.fromfile(path); cfg.merge_from_dict(opts)
    model = init_detector(cfg, checkpoint=ckpt, device='cuda:0')      # MODELS.build(cfg.model) + weights + .eval()
    pipeline = Compose(cfg.test_pipeline)
    model.reparameterize(texts)
    pred = inference_detector(model, image_path, texts, pipeline, max_dets=100, score_thr=0.05)
"""
from __future__ import annotations

import os
import warnings
from typing import List, Optional, Union

import torch

from . import config as _config  # noqa: F401  (fills the registries)
from . import pipeline as _pipeline  # noqa: F401
from .cfgfile import Config
from .detector import InstanceData
from .pipeline import Compose
from .registry import MODELS


def get_test_pipeline_cfg(cfg):
    """``mmdet.utils.get_test_pipeline_cfg``: the test dataloader's pipeline (through dataset wrappers), else the
    top-level ``test_pipeline``."""
    ds = (cfg.get("test_dataloader") or {}).get("dataset") if hasattr(cfg, "get") else None
    while isinstance(ds, dict):
        if "pipeline" in ds:
            return ds["pipeline"]
        ds = ds.get("dataset") or (ds.get("datasets") or [None])[0]
    return cfg.get("test_pipeline")


def load_checkpoint_file(path: str) -> dict:
    """``torch.load(path, map_location='cpu')`` -> the state dict (mmengine wrapper ``{'state_dict': ...}`` or a flat
    dict, as the Uni checkpoints are: generate_proposal.py:1233)."""
    # the safe loader first (the reference calls torch.load with the library default, which is weights_only=True on
    # current torch); mmengine checkpoints that pickle non-tensor metadata need the unsafe one, which executes
    # arbitrary code from the file: that is an explicit opt-in ($WEDETECT_UNSAFE_LOAD=1), never a silent fallback
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except TypeError:                                   # very old torch: no weights_only keyword
        ckpt = torch.load(path, map_location="cpu")
    except Exception as e:
        if os.environ.get("WEDETECT_UNSAFE_LOAD") != "1":
            raise RuntimeError(f"{path} cannot be read by the safe loader (torch.load(weights_only=True)): {e}\n"
                               "If you trust the file, set WEDETECT_UNSAFE_LOAD=1 to unpickle it fully.") from e
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {path}")
    return ckpt["state_dict"] if isinstance(ckpt.get("state_dict"), dict) else ckpt


def init_detector(config: Union[str, os.PathLike, Config], checkpoint: Optional[str] = None, palette: str = "none",
                  device: str = "cuda:0", cfg_options: Optional[dict] = None, tokenizer=None, precision: Optional[str] = None):
    """``mmdet.apis.init_detector``: config (path or Config) -> detector with weights, on ``device``, in eval mode,
    carrying ``.cfg`` and ``.dataset_meta``.  ``tokenizer`` / ``precision`` are this package's additions (a tokenizer
    callable for the text tower when its files are not on disk; "fp32" / "fp16x3")."""
    if isinstance(config, (str, os.PathLike)):
        config = Config.fromfile(config)
    elif not isinstance(config, Config):
        raise TypeError(f"config must be a filename or Config object, but got {type(config)}")
    if cfg_options is not None:
        config.merge_from_dict(cfg_options)
    model_cfg = config.model.to_dict()
    if isinstance(model_cfg.get("backbone"), dict):
        model_cfg["backbone"].pop("init_cfg", None)
    img_scale = config.get("img_scale")
    model = _config.build_detector(model_cfg, img_scale=img_scale, tokenizer=tokenizer, precision=precision)
    if checkpoint is None:
        warnings.simplefilter("once")
        warnings.warn("checkpoint is None: the detector has no weights and cannot run until load_state_dict() is called")
        model.dataset_meta = {"classes": ()}
    else:
        sd = load_checkpoint_file(checkpoint)
        msg = model.load_state_dict(sd, strict=False)
        if msg.unexpected_keys:
            warnings.warn(f"unexpected keys in {checkpoint}: {msg.unexpected_keys[:8]}{' ...' if len(msg.unexpected_keys) > 8 else ''}")
        model.dataset_meta = {"classes": ()}
    if palette != "none":
        model.dataset_meta["palette"] = palette
    model.cfg = config
    model.to(device)
    model.eval()
    return model


def inference_detector(model, image, texts, test_pipeline: Compose, max_dets: int = 100, score_thr: float = 0.3) -> InstanceData:
    """The detection part of the reference demo's ``inference_detector`` (infer_wedetect.py:102-131): pipeline ->
    ``test_step`` -> score filter -> top ``max_dets``; returns the host-side ``InstanceData`` (numpy fields
    ``bboxes`` [n, 4] in original-image pixels, ``scores``, ``labels``).  Drawing is the caller's business."""
    data_info = dict(img_id=0, img_path=image, texts=texts) if isinstance(image, (str, os.PathLike)) else \
        dict(img_id=0, img=image, img_path=None, texts=texts)
    data_info = test_pipeline(data_info)
    data_batch = dict(inputs=data_info["inputs"].unsqueeze(0), data_samples=[data_info["data_samples"]])
    with torch.no_grad():
        output = model.test_step(data_batch)[0]
        pred = output.pred_instances
        pred = pred[pred.scores.float() > score_thr]
    if len(pred.scores) > max_dets:
        indices = pred.scores.float().topk(max_dets)[1]
        pred = pred[indices]
    return pred.cpu().numpy()


def build_model(cfg: dict, **kw):
    return MODELS.build(cfg, **kw)
