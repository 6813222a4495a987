"""Device proposal-recall matching (wd_recall_match) against the reference's goldens and the oracle."""
import numpy as np
import pytest

from tests.util import golden

pytestmark = pytest.mark.gpu


def _fixture():
    fx = golden("recall.npz")
    n = int(fx["count"])
    gts = [None if bool(fx[f"gt{i}_none"]) else fx[f"gt{i}"] for i in range(n)]
    return fx, gts, [fx[f"prop{i}"] for i in range(n)]


def test_eval_recalls_reproduces_reference_goldens():
    from wedetect_amd.evaluate import eval_recalls
    fx, gts, props = _fixture()
    for leg in (False, True):
        got = eval_recalls(gts, props, fx["nums"], fx["thrs"], use_legacy_coordinate=leg)
        assert np.array_equal(got, fx[f"recalls_legacy{int(leg)}"]), f"legacy={leg}"
    assert eval_recalls(gts, props, 100, 0.5).shape == (1, 1)           # scalar parameters like the reference


def test_matched_ious_bit_exact_vs_oracle_on_a_larger_set():
    """300 proposals x up to 40 ground truths per image, 24 images, 3 budgets: the greedy assignment and its fp32
    IoUs equal the numpy restatement element for element (ties included: duplicated boxes)."""
    from oracle import evaluate as oe
    from wedetect_amd.evaluate import matched_ious
    g = np.random.default_rng(77)
    gts, props = [], []
    for i in range(24):
        ng, npr = int(g.integers(0, 41)), int(g.integers(0, 301))
        a = g.uniform(0, 600, (ng, 2)); a = np.concatenate([a, a + g.uniform(5, 250, (ng, 2))], 1).astype(np.float32)
        b = g.uniform(0, 600, (npr, 2)); b = np.concatenate([b, b + g.uniform(5, 250, (npr, 2))], 1).astype(np.float32)
        if ng > 2 and npr > 6:
            b[0], b[5] = a[1], a[1]
            b[3] = a[2] + np.float32(1.5)
        gts.append(a)
        props.append(b)
    nums = np.array([10, 100, 300])
    got = matched_ious(gts, props, nums)
    ref = oe.matched_ious(gts, props, nums)
    assert got.shape == ref.shape and np.array_equal(got, ref)
