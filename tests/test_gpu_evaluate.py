"""Device proposal-recall matching (wd_recall_match) against the reference's goldens and the oracle."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, golden

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
    # images are processed in chunks sized to a scratch budget: tiny budgets (one image per launch, a
    # few images per launch) must give the same array
    for budget in (1, 200_000, 1_000_000):
        assert np.array_equal(matched_ious(gts, props, nums, scratch_budget=budget), ref), budget


def test_retrieval_predictions_match_the_references_lines(tmp_path):
    """retrieval_metric.py:365-377 restated on the CPU vs ``retrieval_predictions`` on a file written by
    ``save_retrieval_file`` (thresholds kept away from any score by construction of the check)."""
    from wedetect_amd import evaluate as E
    g = torch.Generator().manual_seed(17)
    n, r, d, k = 37, 300, 768, 80
    emb = torch.nn.functional.normalize(torch.randn(n, r, d, generator=g), dim=2)
    sc, bi = torch.randn(n, r, generator=g) * 0.2 + 1.5, torch.randn(n, r, generator=g) * 0.3 - 1.0
    counts = torch.randint(0, r + 1, (n,), generator=g, dtype=torch.int32)
    counts[3], counts[5] = 0, r
    text = torch.nn.functional.normalize(torch.randn(k, d, generator=g), dim=1)
    names = [f"class_{i}" for i in range(k)]
    path = str(tmp_path / "synthetic_base.pth")
    E.save_retrieval_file(path, E.retrieval_records(list(range(100, 100 + n)), emb, counts, sc, bi), text)
    pred = E.load_retrieval_file(path)
    ref_scores = torch.zeros(n, k)
    for i, res in enumerate(pred["image_embedding"]):
        if res["embedding"].shape[0] == 0:
            continue                                         # the reference's torch.max would raise on an empty image
        logits = torch.einsum("bw,kw->bk", res["embedding"], pred["text_embedding"])
        logits = torch.sigmoid(logits * res["scale"].exp().unsqueeze(1) + res["bias"].unsqueeze(1))
        ref_scores[i] = torch.max(logits, dim=0)[0]
    got = E.retrieval_scores(pred["image_embedding"], pred["text_embedding"]).cpu()
    assert_close("retrieval scores from file", got, ref_scores, 2e-6, 1e-5)
    v = torch.sort(ref_scores[ref_scores > 0]).values          # a threshold in the widest gap near the median score
    mid = v.numel() // 2
    j = mid - 25 + int(torch.argmax(v[mid - 24:mid + 26] - v[mid - 25:mid + 25]))
    thre = float((v[j] + v[j + 1]) / 2)
    gap = float((ref_scores - thre).abs().min())
    assert gap > 1e-5, "pick another seed: a score sits on the threshold"
    out = E.retrieval_predictions(pred, names, thre)
    ref = {nm: [] for nm in names}
    for i, res in enumerate(pred["image_embedding"]):
        for ids in torch.where(ref_scores[i] > thre)[0].tolist():
            ref[names[ids]].append(res["image_id"])
    assert out == ref and sum(len(v) for v in out.values()) > 0
    results = E.evaluate_retrieval_per_class(out, {nm: set(v[:2]) | {1} for nm, v in ref.items()})
    p_, r_, f_ = E.macro_average(results)
    assert 0.0 <= p_ <= 1.0 and 0.0 <= r_ <= 1.0 and 0.0 <= f_ <= 1.0


def test_device_retrieval_scores_and_single_process_sharded_path():
    """parallel.device_retrieval_scores (the default scorer of class_sharded_retrieval) vs the file-based scorer and
    the reference's lines; without a process group the sharded entry point is the plain scorer."""
    from wedetect_amd import evaluate as E
    from wedetect_amd.parallel import class_sharded_retrieval, device_retrieval_scores
    g = torch.Generator().manual_seed(23)
    n, r, d, k = 9, 300, 768, 203
    emb = torch.nn.functional.normalize(torch.randn(n, r, d, generator=g), dim=2)
    sc, bi = torch.randn(n, r, generator=g) * 0.2 + 1.5, torch.randn(n, r, generator=g) * 0.3 - 1.0
    counts = torch.tensor([300, 0, 1, 17, 300, 250, 3, 64, 128], dtype=torch.int32)
    bank = torch.nn.functional.normalize(torch.randn(k, d, generator=g), dim=1)
    dev = lambda t: t.cuda()
    got = device_retrieval_scores(dev(emb), dev(counts), dev(sc), dev(bi), dev(bank))
    same = class_sharded_retrieval(dev(emb), dev(counts), dev(sc), dev(bi), dev(bank), k)
    assert torch.equal(got, same)
    via_file = E.retrieval_scores(E.retrieval_records(list(range(n)), emb, counts, sc, bi), bank)
    assert torch.equal(got, via_file)
    ref = torch.zeros(n, k)
    for i in range(n):
        c = int(counts[i])
        if c:
            lg = torch.sigmoid(torch.einsum("bw,kw->bk", emb[i, :c], bank) * sc[i, :c].exp().unsqueeze(1) + bi[i, :c].unsqueeze(1))
            ref[i] = lg.max(dim=0)[0]
    assert_close("device retrieval scores", got, ref, 2e-6, 1e-5)
    with pytest.raises(ValueError):
        class_sharded_retrieval(dev(emb), dev(counts), dev(sc), dev(bi), dev(bank[:100]), k)
