"""TEST INFRASTRUCTURE — numpy restatement of OpenCV's 8-bit ``cv2.resize`` for the two interpolation modes the
reference's mmdet test pipeline uses.  Only ``tests/`` may import this module; the product path
(wedetect_amd/pipeline.py + csrc/preprocess.hip) never does.

Reference call site: ``WeDetectKeepRatioResize._resize_img`` (wedetect/datasets/transformers/transforms.py:94-123):
``mmcv.imresize(img, (int(w * ratio), int(h * ratio)), interpolation='area' if ratio < 1 else 'bilinear')`` which is
``cv2.resize(img, size, interpolation=cv2.INTER_AREA | cv2.INTER_LINEAR)`` on a uint8 HxWx3 BGR array.

Third-party algorithm, absent from /root/reference and from this image (``import cv2`` fails; opencv-python is
un-pinned in the reference's README, mmcv 2.1.0 requires opencv-python >= 3): **parity unpinned**.  What is restated is
the published algorithm of OpenCV 4.x ``modules/imgproc/src/resize.cpp`` (the non-IPP, non-"exact" C++ paths: IPP is
skipped for 8-bit linear / area unless ``useIPP_NotExact``), pinned here only against hand-derived vectors
(tests/test_cpu.py) — not against a cv2 run:

``INTER_AREA`` with ``scale = src/dst >= 1`` on both axes
  * both scales integer ("area fast"): each output = box sum of ``sx*sy`` inputs; for 2x2 ``(sum + 2) >> 2``, otherwise
    ``cvRound(sum * (1.f / area))`` (float32 product, round half to even);
  * otherwise: ``computeResizeAreaTab`` per axis — for output d the covered interval ``[d*scale, (d+1)*scale)`` gives up
    to one partial left cell, whole cells with weight ``1/cellWidth`` and one partial right cell (weights are float32,
    partial cells thinner than 1e-3 are dropped, ``cellWidth = min(scale, ssize - d*scale)``) — then, per output row,
    for every contributing source row in order: ``buf[x] = sum_k S[si_k] * alpha_k`` (float32, left to right, separate
    multiply and add), ``sum[x] = beta * buf[x]`` for the first contributing row and ``sum[x] += beta * buf[x]`` after;
    output ``saturate_cast<uchar>(cvRound(sum))``.

``INTER_LINEAR`` (used when the image is scaled UP, ratio > 1): fixed point with 11 fractional bits.
  ``fx = float((dx + 0.5) * scale_x - 0.5)``, ``sx = floor(fx)``, ``fx -= sx``; ``sx < 0 -> (0, 0)``, ``sx >= w - 1 ->
  (w - 1, 0)``; coefficients ``cvRound((1.f - fx) * 2048)``, ``cvRound(fx * 2048)`` as int16; horizontal pass in int32
  (``S[sx] * a0 + S[sx + 1] * a1``, or ``S[sx] * 2048`` from the first column whose ``sx + 1`` leaves the image);
  rows ``clip(sy, 0, h - 1)`` / ``clip(sy + 1, 0, h - 1)`` with ``fy`` NOT zeroed at the border; vertical pass
  ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``.
"""
from __future__ import annotations

import math

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS
DBL_EPSILON = 2.220446049250313e-16


def _cv_round_f32(x: np.ndarray) -> np.ndarray:
    """cvRound on float32 values: round half to even (SSE cvtss2si / lrintf in the default rounding mode)."""
    return np.rint(x.astype(np.float32)).astype(np.int64)


def _sat_u8(v: np.ndarray) -> np.ndarray:
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_scales(src_hw, dst_hw):
    """(scale_x, scale_y, iscale_x, iscale_y, is_area_fast) as cv::hal::resize computes them from dsize."""
    (sh, sw), (dh, dw) = src_hw, dst_hw
    inv_x, inv_y = float(dw) / sw, float(dh) / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    # saturate_cast<int>(double) = cvRound(double) (half to even)
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))
    fast = abs(scale_x - isx) < DBL_EPSILON and abs(scale_y - isy) < DBL_EPSILON
    return scale_x, scale_y, isx, isy, fast


def area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: list of (di, si, alpha float32) in table order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(math.ceil(fsx1)), int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _resize_area_fast(src: np.ndarray, dh: int, dw: int, isx: int, isy: int) -> np.ndarray:
    s = src[: dh * isy, : dw * isx].astype(np.int64)
    s = s.reshape(dh, isy, dw, isx, src.shape[2]).sum(axis=(1, 3))
    if isx == 2 and isy == 2:
        return ((s + 2) >> 2).astype(np.uint8)
    scale = np.float32(1.0) / np.float32(isx * isy)
    return _sat_u8(_cv_round_f32(s.astype(np.float32) * scale))


def _resize_area(src: np.ndarray, dh: int, dw: int, scale_x: float, scale_y: float) -> np.ndarray:
    sh, sw, cn = src.shape
    xtab = area_tab(sw, dw, scale_x)
    ytab = area_tab(sh, dh, scale_y)
    out = np.zeros((dh, dw, cn), np.uint8)
    sf = src.astype(np.float32)
    # per source row: buf[dx] accumulated over the x table in order (float32, mul then add)
    xd = np.array([t[0] for t in xtab]); xs = np.array([t[1] for t in xtab]); xa = np.array([t[2] for t in xtab], np.float32)

    def hrow(sy: int) -> np.ndarray:
        buf = np.zeros((dw, cn), np.float32)
        row = sf[sy]
        # entries of one dx are consecutive; process "k-th entry of every dx" together to keep the per-dx order
        order = np.zeros(len(xtab), np.int64)
        cnt = {}
        for i, d in enumerate(xd):
            order[i] = cnt.get(d, 0)
            cnt[d] = order[i] + 1
        for k in range(int(order.max()) + 1):
            m = order == k
            buf[xd[m]] = (buf[xd[m]] + (row[xs[m]] * xa[m][:, None]).astype(np.float32)).astype(np.float32)
        return buf

    cache = {}
    prev = None
    acc = None
    for (dy, sy, beta) in ytab:
        if sy not in cache:
            cache[sy] = hrow(sy)
        buf = cache[sy]
        if dy != prev:
            if prev is not None:
                out[prev] = _sat_u8(_cv_round_f32(acc))
            acc = (np.float32(beta) * buf).astype(np.float32)
            prev = dy
        else:
            acc = (acc + (np.float32(beta) * buf).astype(np.float32)).astype(np.float32)
    if prev is not None:
        out[prev] = _sat_u8(_cv_round_f32(acc))
    return out


def linear_tab(ssize: int, dsize: int, scale: float):
    """Per output index: (source index, int16 coefficient pair, two_tap flag) of the INTER_LINEAR horizontal pass, and
    the un-clamped source index + coefficients for the vertical use (``clamp=False``)."""
    ofs = np.zeros(dsize, np.int64)
    coef = np.zeros((dsize, 2), np.int64)
    xmax = dsize
    raw = np.zeros(dsize, np.int64)
    rawcoef = np.zeros((dsize, 2), np.int64)
    for dx in range(dsize):
        fx = np.float32((dx + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        fx = np.float32(fx - np.float32(sx))
        raw[dx] = sx
        rawcoef[dx] = (int(np.rint(np.float32((np.float32(1.0) - fx) * np.float32(COEF_SCALE)))),
                       int(np.rint(np.float32(fx * np.float32(COEF_SCALE)))))
        if sx < 0:
            fx, sx = np.float32(0.0), 0
        if sx + 1 >= ssize:
            xmax = min(xmax, dx)
            if sx >= ssize - 1:
                fx, sx = np.float32(0.0), ssize - 1
        ofs[dx] = sx
        coef[dx] = (int(np.rint(np.float32((np.float32(1.0) - fx) * np.float32(COEF_SCALE)))),
                    int(np.rint(np.float32(fx * np.float32(COEF_SCALE)))))
    return ofs, coef, xmax, raw, rawcoef


def _resize_linear(src: np.ndarray, dh: int, dw: int, scale_x: float, scale_y: float) -> np.ndarray:
    sh, sw, cn = src.shape
    xofs, alpha, xmax, _, _ = linear_tab(sw, dw, scale_x)
    _, _, _, yofs, beta = linear_tab(sh, dh, scale_y)
    s = src.astype(np.int64)
    two = np.arange(dw) < xmax
    x1 = np.minimum(xofs + 1, sw - 1)
    hbuf = np.where(two[None, :, None],
                    s[:, xofs] * alpha[:, 0][None, :, None] + s[:, x1] * alpha[:, 1][None, :, None],
                    s[:, xofs] * COEF_SCALE)                                  # [sh, dw, cn] int
    r0 = np.clip(yofs, 0, sh - 1)
    r1 = np.clip(yofs + 1, 0, sh - 1)
    s0, s1 = hbuf[r0], hbuf[r1]
    b0, b1 = beta[:, 0][:, None, None], beta[:, 1][:, None, None]
    v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)                                        # the C code casts with uchar(...)


def cv2_resize_u8(src: np.ndarray, dsize_wh, interpolation: str) -> np.ndarray:
    """``cv2.resize(src, dsize_wh, interpolation=...)`` for uint8 HxWxC, interpolation 'area' or 'bilinear'."""
    if src.dtype != np.uint8 or src.ndim != 3:
        raise TypeError("uint8 HxWxC arrays only")
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    if dw <= 0 or dh <= 0:
        raise ValueError("empty destination")
    sh, sw = src.shape[:2]
    scale_x, scale_y, isx, isy, fast = resize_scales((sh, sw), (dh, dw))
    if interpolation == "area":
        if scale_x >= 1 and scale_y >= 1:
            if fast:
                return _resize_area_fast(src, dh, dw, isx, isy)
            return _resize_area(src, dh, dw, scale_x, scale_y)
        raise NotImplementedError("INTER_AREA with an up-scaled axis (cv2 emulates it with a bilinear variant); the "
                                  "reference only asks for 'area' when both sides shrink")
    if interpolation == "bilinear":
        if fast and isx == 2 and isy == 2:
            return _resize_area_fast(src, dh, dw, 2, 2)                        # cv2 switches to INTER_AREA for exact 2x
        return _resize_linear(src, dh, dw, scale_x, scale_y)
    raise ValueError(f"interpolation {interpolation!r}")


def keep_ratio_resize(img: np.ndarray, scale_wh=(640, 640)) -> np.ndarray:
    """The pixels of WeDetectKeepRatioResize (transforms.py:94-123): ratio = min(max(scale) / max(h, w), min(scale) /
    min(h, w)); size (int(w * ratio), int(h * ratio)); 'area' when shrinking, 'bilinear' when enlarging, untouched at 1."""
    h, w = img.shape[:2]
    ratio = min(max(scale_wh) / max(h, w), min(scale_wh) / min(h, w))
    if ratio == 1:
        return img
    return cv2_resize_u8(img, (int(w * ratio), int(h * ratio)), "area" if ratio < 1 else "bilinear")


def letter_pad(img: np.ndarray, scale_wh=(640, 640), pad_val: int = 114):
    """WeDetectLetterResize with allow_scale_up=False on an image that already fits (transforms.py:180-272): pad to
    (scale_h, scale_w) with top = int(round(pad_h // 2 - 0.1)), left likewise; returns (canvas, pad_param float32)."""
    sh, sw = scale_wh[1], scale_wh[0]
    h, w = img.shape[:2]
    ratio = min(min(sh / h, sw / w), 1.0)
    nh, nw = int(round(h * ratio)), int(round(w * ratio))
    if (nh, nw) != (h, w):
        img = cv2_resize_u8(img, (nw, nh), "bilinear")
    ph, pw = sh - nh, sw - nw
    top, left = int(round(ph // 2 - 0.1)), int(round(pw // 2 - 0.1))
    out = np.full((sh, sw, img.shape[2]), pad_val, np.uint8)
    out[top: top + nh, left: left + nw] = img
    return out, np.array([top, ph - top, left, pw - left], np.float32)
