"""The mmdet test pipeline of config/wedetect_*.py:111-133 with the pixel work on the device.

    test_pipeline = Compose(cfg.test_pipeline)                       (infer_wedetect.py:160)
    data_info = test_pipeline(dict(img_id=0, img_path=path, texts=texts))
    data_batch = dict(inputs=data_info['inputs'].unsqueeze(0), data_samples=[data_info['data_samples']])

Transforms (registered in ``registry.TRANSFORMS`` under the reference's names, same constructor keywords):

  LoadImageFromFile        host: decode the file (PIL; EXIF orientation applied like cv2.imread), BGR uint8 HWC,
                           uploaded once — everything after it stays in HBM
  WeDetectKeepRatioResize  transforms.py:28-123: ratio / target size / ``scale_factor`` computed as the reference does;
                           the resample itself is DEFERRED to the letter step so resize + pad are one kernel
  WeDetectLetterResize     transforms.py:126-328 (allow_scale_up=False, constant pad): ``pad_param``, ``scale_factor``
                           product; launches ``wd_cv_resize_paste_u8`` — cv2's INTER_AREA (integer box fast path or the
                           general float table path) when shrinking, INTER_LINEAR (11-bit fixed point) when enlarging,
                           written straight into the padded canvas
  LoadAnnotations          no-op at inference
  LoadText                 mm_transforms.py:107-135: ``texts`` [[a], [b]] -> [a, b]
  PackDetInputs            ``inputs`` = [3, H, W] uint8 BGR view of the canvas + ``DetDataSample`` with the listed
                           ``meta_keys``

The resize tables are host arithmetic (float64 / float32 exactly as OpenCV computes them) cached per (src, dst)
size; pixels never visit the host after the upload.  cv2 itself is third-party and absent here: see
oracle/cv2_resize.py for the restated algorithm and its "parity unpinned" status.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L
from .detector import DetDataSample
from .registry import TRANSFORMS

COEF_SCALE = 2048
_DBL_EPS = 2.220446049250313e-16


# --------------------------------------------------------------------------------------------------
# host-side table arithmetic (OpenCV resize.cpp: computeResizeAreaTab, the INTER_LINEAR offset loop)
# --------------------------------------------------------------------------------------------------
@lru_cache(maxsize=256)
def _area_table(ssize: int, dsize: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(ranges int32 [dsize, 2], source index int32 [n], weight float32 [n]) for one axis of INTER_AREA."""
    scale = 1.0 / (float(dsize) / ssize)
    ranges = np.zeros((dsize, 2), np.int32)
    idx: List[int] = []
    wts: List[np.float32] = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(math.ceil(f1)), int(math.floor(f2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        start = len(idx)
        if s1 - f1 > 1e-3:
            idx.append(s1 - 1)
            wts.append(np.float32((s1 - f1) / cell))
        for s in range(s1, s2):
            idx.append(s)
            wts.append(np.float32(1.0 / cell))
        if f2 - s2 > 1e-3:
            idx.append(s2)
            wts.append(np.float32(min(min(f2 - s2, 1.0), cell) / cell))
        ranges[d] = (start, len(idx) - start)
    return ranges, np.asarray(idx, np.int32), np.asarray(wts, np.float32)


def _q11(v: np.float32) -> int:
    return int(np.rint(np.float32(v * np.float32(COEF_SCALE))))          # saturate_cast<short>(cvRound(.)): |v| <= 1


@lru_cache(maxsize=256)
def _linear_table(ssize: int, dsize: int, clamp: bool) -> Tuple[np.ndarray, np.ndarray, int]:
    """(coefficient pairs int32 [dsize, 2], source index int32 [dsize], xmax).  ``clamp``: the horizontal form (index
    and fraction clamped at both borders, single tap from xmax on); rows keep the raw index and fraction."""
    scale = 1.0 / (float(dsize) / ssize)
    coef = np.zeros((dsize, 2), np.int32)
    ofs = np.zeros(dsize, np.int32)
    xmax = dsize
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        if clamp:
            if s < 0:
                f, s = np.float32(0.0), 0
            if s + 1 >= ssize:
                xmax = min(xmax, d)
                if s >= ssize - 1:
                    f, s = np.float32(0.0), ssize - 1
        ofs[d] = s
        coef[d] = (_q11(np.float32(1.0) - f), _q11(f))
    return coef, ofs, xmax


def resize_plan(sh: int, sw: int, dh: int, dw: int, interpolation: str) -> dict:
    """Which kernel mode ``cv2.resize((sh, sw) -> (dh, dw), interpolation)`` is, with its host tables (numpy)."""
    if (dh, dw) == (sh, sw):
        return dict(mode=L.CVRESIZE_COPY)
    scale_x, scale_y = 1.0 / (float(dw) / sw), 1.0 / (float(dh) / sh)
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))
    fast = abs(scale_x - isx) < _DBL_EPS and abs(scale_y - isy) < _DBL_EPS
    if interpolation == "bilinear" and fast and isx == 2 and isy == 2:
        interpolation = "area"                                             # resize.cpp: exact 2x bilinear == area
    if interpolation == "area":
        if scale_x < 1 or scale_y < 1:
            raise NotImplementedError("INTER_AREA on an enlarged axis is outside the test pipeline (area is only chosen when shrinking)")
        if fast:
            return dict(mode=L.CVRESIZE_AREA_FAST, p0=isx, p1=isy, p2=float(np.float32(1.0) / np.float32(isx * isy)))
        xa, xi, xw = _area_table(sw, dw)
        ya, yi, yw = _area_table(sh, dh)
        return dict(mode=L.CVRESIZE_AREA, xa=xa, xidx=xi, xw=xw, ya=ya, yidx=yi, yw=yw)
    if interpolation == "bilinear":
        xa, xi, xmax = _linear_table(sw, dw, True)
        ya, yi, _ = _linear_table(sh, dh, False)
        return dict(mode=L.CVRESIZE_LINEAR, xa=xa, xidx=xi, ya=ya, yidx=yi, p0=xmax)
    raise ValueError(f"interpolation {interpolation!r} (the test pipeline uses 'area' and 'bilinear')")


class _PlanCache:
    """Device copies of the resize tables, keyed by (src size, dst size, interpolation)."""

    def __init__(self, device):
        self.dev = device
        self._c: Dict[tuple, dict] = {}

    def get(self, sh, sw, dh, dw, interp) -> dict:
        key = (sh, sw, dh, dw, interp)
        p = self._c.get(key)
        if p is None:
            host = resize_plan(sh, sw, dh, dw, interp)
            p = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(self.dev) if isinstance(v, np.ndarray) else v)
                 for k, v in host.items()}
            if len(self._c) > 128:
                self._c.clear()
            self._c[key] = p
        return p


def cv_resize_pad(src_hwc: torch.Tensor, dh: int, dw: int, interp: str, canvas_hw: Tuple[int, int], top: int, left: int,
                  pad_val: int = 114, swap_rb: bool = False, plans: Optional[_PlanCache] = None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [h, w, 3] device image -> uint8 [H, W, 3] canvas: cv2-style resize to (dh, dw) pasted at (top, left)."""
    if src_hwc.dtype != torch.uint8 or src_hwc.dim() != 3 or src_hwc.shape[2] != 3 or not src_hwc.is_cuda:
        raise L.WedetectHipError("cv_resize_pad: a device uint8 [h, w, 3] image is required")
    src_hwc = src_hwc.contiguous()
    sh, sw = int(src_hwc.shape[0]), int(src_hwc.shape[1])
    plans = plans or _PlanCache(src_hwc.device)
    p = plans.get(sh, sw, dh, dw, interp)
    ch, cw = canvas_hw
    if out is None:
        out = torch.empty(ch, cw, 3, dtype=torch.uint8, device=src_hwc.device)
    L.cv_resize_paste_u8(src_hwc, sh, sw, p["mode"], p.get("xa"), p.get("xidx"), p.get("xw"), p.get("ya"), p.get("yidx"),
                         p.get("yw"), p.get("p0", 0), p.get("p1", 0), p.get("p2", 0.0), out, ch, cw, dh, dw, top, left,
                         pad_val, swap_rb)
    return out


# --------------------------------------------------------------------------------------------------
# transforms
# --------------------------------------------------------------------------------------------------
def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the test pipeline resamples on the GPU (no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


@TRANSFORMS.register_module()
class LoadImageFromFile:
    """``results['img_path']`` (or an already decoded ``results['img']``: HWC uint8 BGR ndarray / tensor) ->
    ``img`` (device uint8 HWC BGR), ``img_shape``, ``ori_shape``."""

    def __init__(self, to_float32: bool = False, color_type: str = "color", imdecode_backend: str = "cv2",
                 file_client_args=None, ignore_empty: bool = False, backend_args=None):
        if to_float32 or color_type != "color":
            raise NotImplementedError("LoadImageFromFile: only color uint8 loading is on the inference path")
        self.ignore_empty = ignore_empty

    def __call__(self, results: dict) -> Optional[dict]:
        img = results.get("img")
        if img is None:
            from PIL import Image, ImageOps
            try:
                with Image.open(results["img_path"]) as im:
                    im = ImageOps.exif_transpose(im).convert("RGB")
                    img = np.asarray(im)[:, :, ::-1]                       # RGB -> BGR (cv2.imread order)
            except Exception:
                if self.ignore_empty:
                    return None
                raise
        if isinstance(img, np.ndarray):
            if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                raise TypeError("img must be uint8 HxWx3")
            img = torch.from_numpy(np.ascontiguousarray(img))
        results["img"] = img.to(_device(), non_blocking=True)
        results["img_shape"] = tuple(int(v) for v in img.shape[:2])
        results["ori_shape"] = tuple(int(v) for v in img.shape[:2])
        return results


def _check_scale(scale):
    if isinstance(scale, (list, tuple)) and len(scale) == 2:
        return tuple(int(v) for v in scale)
    raise TypeError("scale must be a (w, h) pair as in the configs")


@TRANSFORMS.register_module()
class WeDetectKeepRatioResize:
    """transforms.py:28-123.  Records the resample (target size + interpolation) in ``results['_pending_resize']``
    and updates ``img_shape`` / ``scale_factor`` exactly as the reference; pixels move in the next letter step."""

    def __init__(self, scale, keep_ratio: bool = True, **kwargs):
        assert keep_ratio is True
        self.scale = _check_scale(scale)

    def __call__(self, results: dict) -> dict:
        h, w = results["img_shape"][:2]
        ratio = min(max(self.scale) / max(h, w), min(self.scale) / min(h, w))        # :86-89
        if ratio != 1:
            nw, nh = int(w * ratio), int(h * ratio)                                  # :108
            if nw < 1 or nh < 1:
                raise ValueError(f"a {w}x{h} image keeps no pixels at ratio {ratio:.4g}")
            results["_pending_resize"] = (nh, nw, "area" if ratio < 1 else "bilinear")
        else:
            nw, nh = w, h
        results["img_shape"] = (nh, nw)
        results["scale_factor"] = (nw / w, nh / h)                                   # :114-117
        results["scale"] = self.scale
        return results


@TRANSFORMS.register_module()
class WeDetectLetterResize:
    """transforms.py:126-328 for the shipped options (constant pad, no mini-pad / stretch, int pad_param)."""

    def __init__(self, scale, pad_val=dict(img=0, mask=0, seg=255), use_mini_pad: bool = False, stretch_only: bool = False,
                 allow_scale_up: bool = True, half_pad_param: bool = False, **kwargs):
        self.scale = _check_scale(scale)
        if isinstance(pad_val, (int, float)):
            pad_val = dict(img=pad_val, seg=255)
        assert isinstance(pad_val, dict), f"pad_val must be dict, but got {type(pad_val)}"
        if use_mini_pad or stretch_only or half_pad_param:
            raise NotImplementedError("WeDetectLetterResize: use_mini_pad / stretch_only / half_pad_param are not on the test path")
        self.pad_val = int(pad_val.get("img", 0))
        self.allow_scale_up = allow_scale_up
        self._plans: Optional[_PlanCache] = None

    def __call__(self, results: dict) -> dict:
        img = results["img"]
        if "batch_shape" in results:
            sh, sw = (int(v) for v in results["batch_shape"])                        # :186-187
        else:
            sh, sw = self.scale[1], self.scale[0]                                    # :190 (wh -> hw)
        h, w = results["img_shape"][:2]                                              # after the keep-ratio step
        ratio = min(sh / h, sw / w)                                                  # :195
        if not self.allow_scale_up:
            ratio = min(ratio, 1.0)
        nh, nw = int(round(h * ratio)), int(round(w * ratio))                        # :204-205
        pad_h, pad_w = sh - nh, sw - nw
        pending = results.pop("_pending_resize", None)
        if (nh, nw) != (h, w):
            if pending is not None:
                raise NotImplementedError("two successive resamples (keep-ratio then letter resize) are not fused; the "
                                          "shipped pipeline never needs the second one")
            pending = (nh, nw, "bilinear")                                           # :221-225 (Resize default)
        scale_factor = (nw / w, nh / h)                                              # :227-228
        if "scale_factor" in results:                                                # :230-232, 319-326
            o = results["scale_factor"]
            scale_factor = (scale_factor[0] * o[0], scale_factor[1] * o[1])
        results["scale_factor"] = scale_factor
        top, left = int(round(pad_h // 2 - 0.1)), int(round(pad_w // 2 - 0.1))       # :235-236
        dev = img.device
        if self._plans is None or self._plans.dev != dev:
            self._plans = _PlanCache(dev)
        dh, dw, interp = pending if pending is not None else (int(img.shape[0]), int(img.shape[1]), "area")
        results["img"] = cv_resize_pad(img, dh, dw, interp, (sh, sw), top, left, self.pad_val, plans=self._plans)
        results["img_shape"] = (sh, sw, 3)                                           # :260 (image.shape)
        results["pad_param"] = np.array([top, pad_h - top, left, pad_w - left], dtype=np.float32)
        return results


@TRANSFORMS.register_module()
class LoadAnnotations:
    def __init__(self, **kwargs):
        pass

    def __call__(self, results: dict) -> dict:
        return results


@TRANSFORMS.register_module()
class LoadText:
    """mm_transforms.py:107-135: first caption of every class, formatted."""

    def __init__(self, text_path: Optional[str] = None, prompt_format: str = "{}", multi_prompt_flag: str = "/"):
        self.prompt_format = prompt_format
        if text_path is not None:
            import json
            with open(text_path, "r") as f:
                self.class_texts = json.load(f)

    def __call__(self, results: dict) -> dict:
        assert "texts" in results or hasattr(self, "class_texts"), "No texts found in results."
        class_texts = results.get("texts", getattr(self, "class_texts", None))
        texts = []
        for caps in class_texts:
            assert len(caps) > 0
            texts.append(self.prompt_format.format(caps[0]))
        results["texts"] = texts
        return results


@TRANSFORMS.register_module()
class PackDetInputs:
    DEFAULT_KEYS = ("img_id", "img_path", "ori_shape", "img_shape", "scale_factor", "flip", "flip_direction")

    def __init__(self, meta_keys: Sequence[str] = DEFAULT_KEYS):
        self.meta_keys = tuple(meta_keys)

    def __call__(self, results: dict) -> dict:
        img = results["img"]
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img))
        meta = {}
        for k in self.meta_keys:
            if k in results:
                meta[k] = results[k]
        return dict(inputs=img.permute(2, 0, 1), data_samples=DetDataSample(metainfo=meta))


class Compose:
    """``mmengine.dataset.Compose``: builds each dict through the registry, calls them in order; a transform returning
    None ends the chain with None."""

    def __init__(self, transforms: Sequence):
        self.transforms = []
        for t in transforms or []:
            if isinstance(t, dict):
                t = TRANSFORMS.build(t)
            elif not callable(t):
                raise TypeError(f"transform should be a callable object or dict, but got {type(t)}")
            self.transforms.append(t)

    def __call__(self, data: dict) -> Optional[dict]:
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data

    def __repr__(self):
        return "Compose(" + ", ".join(type(t).__name__ for t in self.transforms) + ")"
