"""Shared helpers for the parity tests."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name))


def to_np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def assert_close(name: str, got, ref, atol: float, rtol: float = 0.0):
    """max |got - ref| <= atol + rtol * |ref|, with a diagnostic that locates the worst element."""
    g = to_np(got).astype(np.float64)
    r = to_np(ref).astype(np.float64)
    assert g.shape == r.shape, f"{name}: shape {g.shape} vs {r.shape}"
    if g.size == 0:
        return 0.0
    assert np.all(np.isfinite(g)), f"{name}: non-finite values in result ({np.sum(~np.isfinite(g))} of {g.size})"
    err = np.abs(g - r)
    bound = atol + rtol * np.abs(r)
    worst = np.unravel_index(np.argmax(err - bound), err.shape)
    nbad = int(np.sum(err > bound))
    msg = (f"{name}: {nbad}/{g.size} elements out of tolerance (atol {atol:g}, rtol {rtol:g}); worst at {worst}: "
           f"got {g[worst]:.8g} ref {r[worst]:.8g} |d| {err[worst]:.3g}; max|d| {err.max():.3g}; "
           f"ref rms {np.sqrt(np.mean(r * r)):.3g}")
    assert nbad == 0, msg
    return float(err.max())


def check_checksum(name: str, got_nhwc, fx, prefix: str, atol: float, rtol: float):
    """Compare against a golden checksum record (mean, l2, 64 sampled elements of the NHWC
    flattening) written by tests/golden/make_golden.py."""
    g = to_np(got_nhwc).reshape(-1)
    idx = fx[f"{prefix}.idx"]
    assert_close(f"{name}.samples", g[idx], fx[f"{prefix}.val"], atol, rtol)
    l2 = float(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
    ref_l2 = float(fx[f"{prefix}.l2"])
    assert abs(l2 - ref_l2) <= 1e-4 * ref_l2 + atol, f"{name}: l2 {l2} vs golden {ref_l2}"
    mean = float(np.mean(g.astype(np.float64)))
    assert abs(mean - float(fx[f"{prefix}.mean"])) <= atol + 1e-4 * abs(float(fx[f"{prefix}.mean"])), \
        f"{name}: mean {mean} vs golden {float(fx[f'{prefix}.mean'])}"
