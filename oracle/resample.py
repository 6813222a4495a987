"""TEST INFRASTRUCTURE — CPU restatement of Pillow's 8-bit antialiased BILINEAR resize
(third-party code behind generate_proposal.py:56 ``img.resize(new_unpad, Image.Resampling.BILINEAR)``
and transforms.py's keep-ratio resize; Pillow src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc).  Pillow itself is importable
in this image, so parity is PINNED: tests/test_cpu.py checks this restatement bit for bit against
``PIL.Image.resize`` (Pillow 12.2.0) over up- and down-scaling sizes, and the GPU kernels against both.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size: int, out_size: int):
    """(bounds [out, 2] int32, weights [out, ksize] int32), vectorised over the output axis."""
    scale = float(np.float32(in_size)) / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    arg = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where((arg < 1.0) & (x < n[:, None]), 1.0 - arg, 0.0)
    ww = np.zeros(out_size)
    for j in range(ksize):                       # Pillow accumulates the normaliser left to right
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    q = w * float(1 << PRECISION_BITS)
    ik = np.where(q < 0, np.trunc(-0.5 + q), np.trunc(0.5 + q)).astype(np.int32)
    return np.stack([xmin, n], axis=1).astype(np.int32), ik


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for o in range(bounds.shape[0]):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.tensordot(kk[o, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """HWC uint8 -> [new_h, new_w, C] uint8; horizontal pass first, uint8 between the passes."""
    h, w = img.shape[:2]
    bh, kh = coeffs(w, new_w)
    bv, kv = coeffs(h, new_h)
    return _pass(_pass(img, bh, kh, 1), bv, kv, 0)


def letterbox_u8(img: np.ndarray, new_shape=(640, 640), fill=(114, 114, 114)):
    """generate_proposal.py:17-82 on an HWC uint8 array: (canvas, ratio, (dw/2, dh/2))."""
    h, w = img.shape[:2]
    tw, th = new_shape[1], new_shape[0]
    r = min(tw / w, th / h)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = tw - nw, th - nh
    canvas = np.empty((th, tw, 3), np.uint8)
    canvas[...] = np.asarray(fill, np.uint8)
    canvas[dh // 2:dh // 2 + nh, dw // 2:dw // 2 + nw] = resize_bilinear_u8(img, nw, nh)
    return canvas, r, (dw / 2, dh / 2)
