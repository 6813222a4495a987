"""Device-side letterbox: keep-ratio bilinear resize + centred pad of
uint8 RGB images, bit-exact with the reference's host path
``img.resize(new_unpad, Image.Resampling.BILINEAR)`` + paste on a 114-grey canvas
(generate_proposal.py:17-82; mmdet path: transforms.py:94-123).

Pillow's resize is a separable, antialiased convolution in 8-bit fixed point
(src/libImaging/Resample.c): per output coordinate a window ``[xmin, xmin+n)`` of input
samples and normalised triangle-filter weights, converted to int32 with 22 fractional bits;
each pass accumulates ``sum(pixel * k) + 2^21`` in int32, shifts right by 22 and clamps to
[0, 255]; the horizontal pass runs first and its uint8 result feeds the vertical pass.  The
weight tables are computed here on the host exactly as Pillow does (float64, same operation
order); the two integer passes and the paste run on the GPU (csrc/preprocess.hip), so the host
no longer touches pixels: it uploads the decoded image once."""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=512)
def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the BILINEAR filter over the
    whole axis: returns (bounds int32 [out, 2] = (first sample, count), weights int32 [out, ksize])."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size      # box = (0, in_size) held as C floats
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        ww = 0.0
        for x in range(n):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(n):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, n)
    q = kk * float(1 << PRECISION_BITS)
    ik = np.where(q < 0, np.trunc(-0.5 + q), np.trunc(0.5 + q)).astype(np.int32)
    return bounds, ik


def letterbox_geometry(w: int, h: int, new_shape=(640, 640)):
    """(new_w, new_h, left, top, ratio, (dw/2, dh/2)) as generate_proposal.py:44-78 computes them."""
    tw, th = new_shape[1], new_shape[0]
    r = min(tw / w, th / h)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = tw - nw, th - nh
    return nw, nh, dw // 2, dh // 2, r, (dw / 2, dh / 2)


def mmdet_test_geometry(ori_h: int, ori_w: int, scale=(640, 640)) -> dict:
    """Shapes and bookkeeping of the mmdet test pipeline of config/wedetect_*.py:111-118 —
    ``WeDetectKeepRatioResize(scale)`` then ``WeDetectLetterResize(scale, allow_scale_up=False, pad_val=114)`` —
    for one ``ori_h x ori_w`` image, exactly as transforms.py:62-123 and 180-272 / 319-330 compute them:

      resized_shape   (h, w) after the keep-ratio resize: ``int(w * ratio)`` (truncated), ratio =
                      min(max(scale) / max(h, w), min(scale) / min(h, w)); unchanged when ratio == 1
      no_pad_shape    (h, w) after the letter resize: ``int(round(. * ratio2))``, ratio2 = min(fit, 1.0) — the keep-ratio
                      step scales small images UP to fit; only this second step never scales up
      pad_param       float32 [top, bottom, left, right], top = int(round(pad_h // 2 - 0.1))
      scale_factor    (w, h) python floats: product of the two transforms' measured size ratios
      img_shape       (scale_h, scale_w)

    ``pad_param`` and ``scale_factor`` are what ``YOLOWorldDetector.predict`` needs to map boxes back to the
    original image (yolo_world_head.py:728-746).  Pixels are host work in the reference (cv2 area / bilinear
    through mmcv.imresize, absent here) and are not reproduced by this function."""
    if not (isinstance(scale, (tuple, list)) and len(scale) == 2):
        raise TypeError("scale must be a (w, h) pair as in the configs")
    scale = tuple(int(v) for v in scale)
    h, w = int(ori_h), int(ori_w)
    if h <= 0 or w <= 0:
        raise ValueError("empty image")
    ratio = min(max(scale) / max(h, w), min(scale) / min(h, w))                      # :86-89
    h1, w1 = (int(h * ratio), int(w * ratio)) if ratio != 1 else (h, w)              # :104-112
    sf1 = (w1 / w, h1 / h)                                                           # :114-117
    sh, sw = scale[::-1]                                                             # :190 (wh -> hw)
    ratio2 = min(min(sh / h1, sw / w1), 1.0)                                         # :195-199 (allow_scale_up=False)
    nh, nw = int(round(h1 * ratio2)), int(round(w1 * ratio2))                        # :204-205
    pad_h, pad_w = sh - nh, sw - nw                                                  # :208-210
    sf2 = (nw / w1, nh / h1)                                                         # :230-231
    top, left = int(round(pad_h // 2 - 0.1)), int(round(pad_w // 2 - 0.1))           # :238-239
    pad_param = np.array([top, pad_h - top, left, pad_w - left], dtype=np.float32)   # :240-245, 271-272
    return dict(resized_shape=(h1, w1), no_pad_shape=(nh, nw), pad_param=pad_param,
                scale_factor=(sf2[0] * sf1[0], sf2[1] * sf1[1]), img_shape=(sh, sw))  # :321-326


class DeviceLetterbox:
    """Letterboxes a list of RGB uint8 images into one [B, H, W, 3] device canvas."""

    def __init__(self, new_shape=(640, 640), fill=(114, 114, 114), device="cuda"):
        self.new_shape = tuple(new_shape)
        self.fill = tuple(int(v) for v in fill)
        self.dev = torch.device(device)
        self._tables = {}                     # (in, out) -> (bounds, weights) on device
        self._tmp = None

    def _table(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            b, k = resample_coeffs(in_size, out_size)
            t = (torch.from_numpy(b).to(self.dev), torch.from_numpy(np.ascontiguousarray(k)).to(self.dev), k.shape[1])
            if len(self._tables) > 256:
                self._tables.clear()
            self._tables[key] = t
        return t

    def __call__(self, images: Sequence, out: torch.Tensor = None):
        """``images``: PIL images, HWC uint8 numpy arrays or HWC uint8 tensors (host or device).
        Returns (canvas uint8 [B, H, W, 3] on the device, ratios, [(dw/2, dh/2)])."""
        th, tw = self.new_shape
        b = len(images)
        if out is None:
            out = torch.empty(b, th, tw, 3, dtype=torch.uint8, device=self.dev)
        elif tuple(out.shape) != (b, th, tw, 3) or out.dtype != torch.uint8 or not out.is_cuda:
            raise L.WedetectHipError(f"out must be a device uint8 [{b},{th},{tw},3] tensor")
        ratios, pads = [], []
        for i, img in enumerate(images):
            src = _as_device_u8(img, self.dev)
            h, w = int(src.shape[0]), int(src.shape[1])
            nw, nh, left, top, r, pad = letterbox_geometry(w, h, self.new_shape)
            if nw < 1 or nh < 1:
                raise L.WedetectHipError(f"image {i}: {w}x{h} letterboxes to an empty {nw}x{nh} image")
            bh, kh, ksh = self._table(w, nw)
            bv, kv, ksv = self._table(h, nh)
            need = h * nw * 3
            if self._tmp is None or self._tmp.numel() < need:
                self._tmp = torch.empty(need, dtype=torch.uint8, device=self.dev)
            L.letterbox_u8(src, h, w, bh, kh, ksh, bv, kv, ksv, self._tmp, out[i], th, tw, nw, nh, left, top, self.fill)
            ratios.append(r)
            pads.append(pad)
        return out, ratios, pads


class DeviceTestPipeline:
    """The mmdet test pipeline's image side for callers without cv2 / mmcv: shapes, ``scale_factor`` and ``pad_param``
    exactly as ``WeDetectKeepRatioResize`` + ``WeDetectLetterResize`` produce them (``mmdet_test_geometry``, pinned to
    the reference's code), pixels resampled on the device with the antialiased bilinear filter of ``wd_letterbox_u8`` in
    ONE pass from the original to the final unpadded size.  The reference resamples on the host with cv2 (area when
    shrinking, bilinear otherwise, in up to two passes): geometry and metadata are identical, pixel values are not —
    use the reference's own pipeline upstream of ``YOLOWorldDetector.predict`` when bit-equal inputs matter.
    Returns BGR->RGB-agnostic uint8 canvases: channels are kept in the order they come in."""

    def __init__(self, scale=(640, 640), pad_val: int = 114, device="cuda"):
        self.scale = tuple(int(v) for v in scale)
        self._lb = DeviceLetterbox(new_shape=self.scale[::-1], fill=(pad_val,) * 3, device=device)

    def __call__(self, images: Sequence):
        """-> (canvas uint8 [B, H, W, 3] on the device, [metainfo dict per image: ori_shape, img_shape, scale_factor,
        pad_param] — the keys ``PackDetInputs`` forwards, config/wedetect_base.py:121-129)."""
        lb = self._lb
        th, tw = lb.new_shape
        out = torch.empty(len(images), th, tw, 3, dtype=torch.uint8, device=lb.dev)
        metas = []
        for i, img in enumerate(images):
            src = _as_device_u8(img, lb.dev)
            h, w = int(src.shape[0]), int(src.shape[1])
            geo = mmdet_test_geometry(h, w, self.scale)
            nh, nw = geo["no_pad_shape"]
            top, left = int(geo["pad_param"][0]), int(geo["pad_param"][2])
            bh, kh, ksh = lb._table(w, nw)
            bv, kv, ksv = lb._table(h, nh)
            need = h * nw * 3
            if lb._tmp is None or lb._tmp.numel() < need:
                lb._tmp = torch.empty(need, dtype=torch.uint8, device=lb.dev)
            L.letterbox_u8(src, h, w, bh, kh, ksh, bv, kv, ksv, lb._tmp, out[i], th, tw, nw, nh, left, top, lb.fill)
            metas.append(dict(ori_shape=(h, w), img_shape=geo["img_shape"], scale_factor=geo["scale_factor"],
                              pad_param=geo["pad_param"]))
        return out, metas


def _as_device_u8(img, dev) -> torch.Tensor:
    if isinstance(img, torch.Tensor):
        t = img
    else:
        a = np.asarray(img)                   # PIL image or ndarray
        if a.ndim != 3 or a.shape[2] != 3:
            raise L.WedetectHipError("images must be RGB, HWC")
        if not a.flags.writeable or not a.flags.c_contiguous:
            a = np.array(a, order="C")        # PIL hands out read-only views; torch wants a writable buffer
        t = torch.from_numpy(a)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise L.WedetectHipError("images must be uint8 HWC RGB")
    return t.to(dev, non_blocking=True).contiguous()
