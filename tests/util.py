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


# ------------------------------------------------------------------------------------------ index parity
PARITY_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_r05.jsonl")
# Reference-generated cases whose kept lists are RECORDED instead of asserted exact.  Round 4 demoted every 640 x 640 golden
# of round 1 by image size; round 5 turns that round: every golden is asserted (a tie-run permutation inside the
# reference's own near-tie scores is the one allowance), and a case is only ever listed here — by NAME, with the observation
# that put it here — after it actually failed with the reference's effective margins inside the measured noise.
# Empty: all of them still reproduce exactly (profiles/r04_parity.jsonl, profiles/r05_parity.jsonl).
KNOWN_NOISE_LEVEL_CASES: dict = {}
RELAXATIONS = []          # (case, kind) of every comparison in this process that was not position-by-position identical
REPORTED = []             # (case, exact) of the comparisons run with assert_exact=False: noise-level references, recorded not asserted


def compare_kept_lists(name, got_anchor, got_label, got_score, ref_anchor, ref_label, ref_score, margins=None,
                       score_tol: float = 1e-3, got_boxes=None, ref_boxes=None, assert_exact: bool = True, eff_margins=None,
                       allow=("tie_run", "cut_swap")):
    """Index parity of a kept-detection list against the reference's (north_star: "box indices/classes bit-exact").

    The assertion is EXACTNESS: the (anchor, class) lists must be identical, position by position, and the scores within
    ``score_tol``.  One relaxation exists and is never silent: rows whose REFERENCE scores lie within ``eps`` of each other
    (eps = 4 x the score difference measured in this very comparison, floor 1e-6 — fp32 summation noise between a CPU run
    and the device) form a "tie run" whose members may appear in any order; it is only considered when the reference's own
    ``kept_gap`` margin says such a run exists (kept_gap < eps), it is printed, appended to RELAXATIONS and to the parity
    log, and ``assert_no_relaxations`` lets a test refuse it altogether.  Anything else fails, with the reference's four
    decision margins ([min |IoU - thr| on the boxes NMS compared, min kept-vs-suppressed score gap, min kept gap, gap at
    the nms_pre cut], recorded by the golden generator) in the message: margins below the measured noise mean the
    reference's own decision was not reproducible by ANY second implementation — that is reported, not waved through.

    Round 4.  (1) ``eff_margins``: the reference's margins restricted to the decisions that can reach the output rows
    (oracle.postprocess.effective_margins); when given they, not the all-candidates ``margins``, say which reference
    decisions lie within the measured noise.  (2) A second counted relaxation, "cut_swap": MEMBERSHIP differences confined to
    rows whose scores lie within eps of the LOWEST kept reference score — the boundary of the max_per_img cut (and of the
    nms_pre cut when NMS runs out of candidates): two candidates closer than the noise trade places across the cut.  Printed,
    appended to RELAXATIONS and the parity log like a tie run.  (3) ``assert_exact=False``: record only (REPORTED, the
    parity log) — for the round-3 goldens, whose margins sit at the noise level and which stay as reported-not-asserted
    cases next to the margin-robust ones; scores are still held to ``score_tol`` and the overlap to 0.97.
    Round 5.  ``allow``: which counted relaxations an asserted comparison may take — the reference-generated goldens pass
    ("tie_run",): a membership trade across the cut fails them.
    Returns (rows of got, rows of ref) of the common detections and appends one JSON line to gpurun_out/parity_r05.jsonl."""
    import json
    ga, gl, gs = (np.asarray(to_np(x)) for x in (got_anchor, got_label, got_score))
    ra, rl, rs = (np.asarray(x) for x in (ref_anchor, ref_label, ref_score))
    assert ga.shape[0] == ra.shape[0], f"{name}: kept {ga.shape[0]} vs reference {ra.shape[0]}"
    n = ra.shape[0]
    assert_close(f"{name} sorted scores", gs, rs, score_tol, 0)
    noise = float(np.max(np.abs(gs.astype(np.float64) - rs.astype(np.float64)))) if n else 0.0
    eps = max(4.0 * noise, 1e-6)
    exact = bool(np.array_equal(ga, ra) and np.array_equal(gl, rl))
    got = list(zip(ga.tolist(), gl.tolist()))
    want = list(zip(ra.tolist(), rl.tolist()))
    overlap = len(set(got) & set(want)) / max(1, n)
    # tie runs of the reference list
    run_exact, runs, longest = True, 0, 1
    i = 0
    rs64 = rs.astype(np.float64)
    while i < n:
        j = i
        while j + 1 < n and rs64[j] - rs64[j + 1] < eps:
            j += 1
        runs += 1
        longest = max(longest, j - i + 1)
        if sorted(got[i:j + 1]) != sorted(want[i:j + 1]):
            run_exact = False
        i = j + 1
    gi = {k: j for j, k in enumerate(want)}
    rows = [(j, gi[k]) for j, k in enumerate(got) if k in gi]
    jj = np.asarray([r[0] for r in rows], dtype=np.int64)
    gg = np.asarray([r[1] for r in rows], dtype=np.int64)
    box_noise, min_side = 0.0, 1.0
    if got_boxes is not None and ref_boxes is not None and len(rows):
        gb, rb = np.asarray(to_np(got_boxes))[jj].astype(np.float64), np.asarray(ref_boxes)[gg].astype(np.float64)
        box_noise = float(np.max(np.abs(gb - rb)))
        side = np.minimum(rb[:, 2] - rb[:, 0], rb[:, 3] - rb[:, 1])
        min_side = float(max(1.0, np.min(side)))
    iou_eps = 8.0 * box_noise / min_side + 1e-7
    within_noise = []
    kept_g = 0.0
    use_m = eff_margins if eff_margins is not None else margins
    if use_m is not None:
        iou_m, pair_g, kept_g, cut_g = (float(v) for v in use_m)
        within_noise = [k for k, bad in (("iou_margin", iou_m <= iou_eps), ("pair_gap", pair_g <= eps), ("kept_gap", kept_g <= eps),
                                         ("cut_gap", cut_g <= eps)) if bad]
    relaxation = "none"
    if not exact and run_exact and (use_m is None or kept_g < eps):
        relaxation = "tie_run"
    elif not exact and n:
        # cut_swap: every row that is in one list only sits within eps of the lowest kept score, and the common rows are in
        # the same order up to tie runs
        sg, sw = set(got), set(want)
        only_g = [j for j, k in enumerate(got) if k not in sw]
        only_w = [j for j, k in enumerate(want) if k not in sg]
        floor = float(rs64[-1])
        near = (all(float(gs[j]) - floor < eps for j in only_g) and all(rs64[j] - floor < eps for j in only_w))
        if (only_g or only_w) and len(only_g) == len(only_w) and near:
            cg = [k for k in got if k in sw]
            cw = [k for k in want if k in sg]
            sc_w = {k: rs64[j] for j, k in enumerate(want)}
            ok, i = True, 0
            while i < len(cw):                                    # same order up to runs of reference scores closer than eps
                j = i
                while j + 1 < len(cw) and sc_w[cw[j]] - sc_w[cw[j + 1]] < eps:
                    j += 1
                ok = ok and sorted(cg[i:j + 1]) == sorted(cw[i:j + 1])
                i = j + 1
            if ok:
                relaxation = "cut_swap"
    if relaxation != "none" and assert_exact:
        RELAXATIONS.append((name, relaxation))
    if not assert_exact:
        REPORTED.append((name, exact))
    rec = dict(case=name, kept=int(n), exact=exact, asserted=bool(assert_exact), relaxation=relaxation, overlap=round(overlap, 4),
               score_noise=noise, eps=eps, tie_runs=runs, longest_run=longest, box_noise=box_noise, iou_eps=iou_eps,
               margins=[float(v) for v in margins] if margins is not None else None,
               eff_margins=[float(v) for v in eff_margins] if eff_margins is not None else None,
               reference_margins_within_noise=within_noise)
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        with open(PARITY_LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    print(f"[parity] {rec}")
    if relaxation != "none":
        print(f"[parity] RELAXATION #{len(RELAXATIONS)} ({relaxation}) granted to {name}: rows inside reference tie runs (score gap < {eps:.2e}) are permuted")
    if not assert_exact:
        assert overlap >= 0.97, f"{name} (reported, not asserted exact): overlap {overlap:.4f} with the reference list"
        return jj, gg
    assert exact or relaxation in tuple(allow), (
        f"{name}: kept (anchor, class) list differs from the reference (overlap {overlap:.4f}, score noise {noise:.2e}, "
        f"margins [iou, pair, kept, cut] {margins}, effective {eff_margins}; reference decisions within the measured noise: "
        f"{within_noise or 'none'})")
    return jj, gg


def assert_no_relaxations(prefix: str = "", allow_tie_runs: bool = False) -> None:
    """For tests whose asserted comparisons must need no relaxation (the BASELINE configurations).  ``allow_tie_runs``: a
    tie-run permutation — two OUTPUT ROWS whose reference scores are closer than the measured noise swap places; it is only
    ever granted when the reference's own (effective) kept-row gap is below that noise — is tolerated and printed; a
    membership change (cut_swap) never is.  The margin-robust goldens have decision margins >= 2e-5 but kept-row gaps of
    3e-6 ... 8e-6 (300 sigmoid scores of one image are nearly continuous; best of 120 / 80 seeds), i.e. 2 - 6 x the measured
    score noise: row order inside such a pair is not something a second implementation can be held to."""
    hit = [r for r in RELAXATIONS if r[0].startswith(prefix) and not (allow_tie_runs and r[1] == "tie_run")]
    ties = [r for r in RELAXATIONS if r[0].startswith(prefix) and r[1] == "tie_run"]
    if allow_tie_runs and ties:
        print(f"[parity] {len(ties)} tie-run permutation(s) tolerated under {prefix!r}: {ties}")
    assert not hit, f"index parity needed {len(hit)} relaxation(s): {hit}"
