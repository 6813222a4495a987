"""Operator-surface parity (GPU): the drop-in detector classes against the CPU oracle, plus the
larger BASELINE configurations (LVIS-size bank, big retrieval bank)."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, to_np

pytestmark = pytest.mark.gpu


def _images():
    from PIL import Image
    g = np.random.default_rng(3)
    return [Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in ((120, 200), (128, 64), (90, 90))]


BOTH = pytest.mark.parametrize("precision", ["fp32", "fp16x3"])


@BOTH
def test_simple_yolo_world_detector_matches_oracle_with_letterbox(precision):
    """generate_proposal.py:1082-1117 end to end: PIL letterbox -> tower -> head_predict -> un-letterbox."""
    from oracle import postprocess as opp
    from oracle import ref_cpu as orc
    from wedetect_amd import weights as W
    from wedetect_amd.arch import HD, get_arch
    from wedetect_amd.detector import SimpleYOLOWorldDetector, letterbox
    sd_np = W.make_state_dict("nano", num_prompts=32)
    model = SimpleYOLOWorldDetector("nano", prompt_dim=768, num_prompts=32, num_proposals=100, precision=precision)
    msg = model.load_state_dict({k: torch.from_numpy(v) for k, v in W.to_uni_keys(sd_np).items()}, strict=False)
    assert not msg.missing_keys
    model = model.cuda()
    model.eval()
    imgs = _images()
    outs = model(imgs)
    assert len(outs) == 3
    sd = orc.to_torch(sd_np)
    ls = np.asarray([sd[HD + f"cls_contrasts.{l}.logit_scale"].item() for l in range(3)], np.float32)
    cb = np.asarray([sd[HD + f"cls_contrasts.{l}.bias"].item() for l in range(3)], np.float32)
    for img, out in zip(imgs, outs):
        lb, ratio, (dw, dh) = letterbox(img, (128, 128))
        x = np.asarray(lb, dtype=np.uint8)[None]
        with torch.no_grad():
            _, p = orc.forward_features(sd, get_arch("nano"), x)
            flat = orc.head_flat(sd, p, sd["embeddings"], normalize_text=False)
        o = opp.uni_predict_image(flat["boxes"][0].numpy(), flat["embed"][0].numpy(), flat["scores"][0].numpy(),
                                  flat["level_of"].numpy(), ls, cb, num_proposals=100)
        n = o["scores"].shape[0]
        assert out["scores"].shape[0] == n and out["labels"].dtype == torch.int64
        assert_close("uni scores", out["scores"], o["scores"], 1e-3)
        same = np.mean(to_np(out["labels"]) == o["labels"])
        assert same > 0.97, f"labels agree on {same:.3f}"
        ref_boxes = opp.unletterbox(o["bboxes"], (dw, dh), ratio, (img.size[1], img.size[0]))
        ok = to_np(out["labels"]) == o["labels"]
        assert_close("uni boxes (original pixels)", to_np(out["bboxes"])[ok], ref_boxes[ok], 5e-2, 1e-5)
        assert_close("uni embeddings", to_np(out["embeddings"])[ok], o["embeddings"][ok], 1e-3, 1e-3)
        assert_close("scales", to_np(out["scales"])[ok], o["scales"][ok], 0)
        assert_close("bias", to_np(out["bias"])[ok], o["bias"][ok], 0)
        assert float(out["bboxes"].min()) >= 0 and float(out["bboxes"][:, 0::2].max()) <= img.size[0]


@BOTH
def test_yolo_world_detector_test_step_matches_oracle(precision):
    """infer_wedetect.py:102-131: BGR CHW uint8 inputs + data samples with letterbox metadata."""
    from oracle import postprocess as opp
    from oracle import ref_cpu as orc
    from wedetect_amd import weights as W
    from wedetect_amd.arch import get_arch
    from wedetect_amd.detector import DetDataSample, YOLOWorldDetector
    sd_np = W.make_state_dict("nano")
    k = 81                                                       # 80 classes + the blank the demo appends
    bank = W.make_text_bank(k) * np.float32(2.5)
    model = YOLOWorldDetector("nano", test_cfg=dict(max_per_img=50), max_classes=k, precision=precision)
    model.load_state_dict({"state_dict": {n: torch.from_numpy(v) for n, v in sd_np.items()}})
    model.cuda().eval()
    with pytest.raises(RuntimeError):
        model.test_step(dict(inputs=[torch.zeros(3, 128, 128, dtype=torch.uint8)], data_samples=[DetDataSample()]))
    model.set_text_embeddings(torch.from_numpy(bank))
    rgb = W.make_images(2, 128, 128, seed=77)
    bgr_chw = [torch.from_numpy(np.ascontiguousarray(im[..., ::-1].transpose(2, 0, 1))) for im in rgb]
    metas = [dict(ori_shape=(200, 256), scale_factor=(0.5, 0.5), pad_param=np.array([14., 14., 0., 0.])),
             dict(ori_shape=(128, 100), scale_factor=(1.0, 1.0), pad_param=np.array([0., 0., 14., 14.]))]
    samples = [DetDataSample(metainfo=m) for m in metas]
    res = model.test_step(dict(inputs=bgr_chw, data_samples=samples))
    sd = orc.to_torch(sd_np)
    with torch.no_grad():
        _, p = orc.forward_features(sd, get_arch("nano"), rgb)
        flat = orc.head_flat(sd, p, torch.from_numpy(bank), normalize_text=True)
    for i, (m, r) in enumerate(zip(metas, res)):
        o = opp.mmdet_predict_image(flat["boxes"][i].numpy(), flat["scores"][i].numpy(), m["pad_param"],
                                    m["scale_factor"], m["ori_shape"], max_per_img=50)
        pi = r.pred_instances
        assert len(pi) == o["scores"].shape[0] == 50
        assert_close("det scores", pi.scores, o["scores"], 1e-3)
        ok = to_np(pi.labels) == o["labels"]
        assert ok.mean() > 0.95
        assert_close("det boxes", to_np(pi.bboxes)[ok], o["bboxes"][ok], 5e-2, 1e-5)
        keep = pi[pi.scores > float(np.median(o["scores"]))]
        assert 0 < len(keep) <= 50 and keep.cpu().numpy()["bboxes"].shape[1] == 4


@BOTH
def test_large_lvis_bank_config3_small_input(precision):
    """WeDetect-Large tower with a 1203-class bank (BASELINE configs[2]) at 128x128 so the CPU
    oracle stays quick: scores parity and exact post-process on >400k candidates per image."""
    from oracle import postprocess as opp
    from oracle import ref_cpu as orc
    from wedetect_amd import weights as W
    from wedetect_amd.arch import get_arch
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    arch, b, hw, k = "large", 2, 128, 1203
    sd_np = W.make_state_dict(arch)
    tower = ImageTower(arch, pack(sd_np, arch), b, hw, hw, max_classes=k, precision=precision)
    imgs = W.make_images(b, hw, hw)
    bank = W.make_text_bank(k)
    sd = orc.to_torch(sd_np)
    with torch.no_grad():
        _, p = orc.forward_features(sd, get_arch(arch), imgs)
        flat = orc.head_flat(sd, p, torch.from_numpy(bank), normalize_text=True)
    meta = tower.identity_meta()
    meta[:, 7] = 1.0
    res = tower.detect(torch.from_numpy(imgs).cuda(), torch.from_numpy(bank).cuda(), meta, normalize_text=True,
                       score_thr=0.001)
    scores, boxes = to_np(tower.scores.view(-1)[: b * tower.ntot * k].view(b, tower.ntot, k)), to_np(tower.boxes)
    assert_close("large scores K=1203", scores, flat["scores"], 1e-3)
    for i in range(b):
        o = opp.mmdet_predict_image(boxes[i], scores[i], None, (1.0, 1.0), (hw, hw))
        n = int(res["count"][i])
        assert n == o["scores"].shape[0]
        assert np.array_equal(to_np(res["anchors"][i, :n]), o["anchors"])
        assert np.array_equal(to_np(res["labels"][i, :n]), o["labels"])
        assert np.array_equal(to_np(res["bboxes"][i, :n]), o["bboxes"])


def test_retrieval_big_bank_properties():
    """configs[4]-style retrieval: 300 regions x 200k-class bank per image without materialising
    logits.  Checked against chunked fp64 math on a sample of classes, and for the class-shard
    identity: scoring bank shards separately == scoring the whole bank."""
    from wedetect_amd import lib as L
    from wedetect_amd import weights as W
    from wedetect_amd.parallel import shard_range
    n_img, rows, k = 2, 300, 200_000
    g = torch.Generator(device="cuda").manual_seed(5)
    e = torch.randn(n_img, rows, 768, device="cuda", generator=g) * 1.4
    t = torch.nn.functional.normalize(torch.randn(k, 768, device="cuda", generator=g), dim=-1)
    scale = torch.full((n_img, rows), -0.35, device="cuda")
    bias = torch.full((n_img, rows), -2.6, device="cuda")
    cnt = torch.tensor([300, 211], dtype=torch.int32, device="cuda")
    out = torch.empty(n_img, k, device="cuda")
    L.retrieval_max(e, t, scale, bias, cnt, out, n_img, rows, k, 768)
    idx = torch.randint(0, k, (512,), device="cuda", generator=g)
    for i in range(n_img):
        lg = e[i, : int(cnt[i])].double() @ t[idx].double().T
        ref = torch.sigmoid(lg * float(np.exp(np.float32(-0.35))) - 2.6).max(dim=0)[0]
        assert_close(f"retrieval img{i} sampled classes", out[i, idx], ref, 2e-6, 1e-5)
    parts = []
    for r in range(8):
        sr = shard_range(k, 8, r)
        o = torch.empty(n_img, len(sr), device="cuda")
        L.retrieval_max(e, t[sr.start:sr.stop].contiguous(), scale, bias, cnt, o, n_img, rows, len(sr), 768)
        parts.append(o)
    assert torch.equal(torch.cat(parts, dim=1), out), "class-sharded scoring must equal whole-bank scoring bit for bit"


def test_fp16x3_range_guard_falls_back_to_fp32():
    """a checkpoint whose GELU hidden activations exceed the fp16 range (one pwconv1 scaled by 1e5,
    undone in the following pwconv2, so the network function is unchanged) turns into inf in the fp16 hi halves.  The
    top-k kernel reports non-finite score rows (count -1), the tower switches to the fp32 MFMA kernels, re-runs, warns,
    and the results equal a tower built in fp32 from the start; an unaffected checkpoint never trips the guard."""
    from wedetect_amd import weights as W
    from wedetect_amd.detector import DetDataSample, YOLOWorldDetector
    sd = W.make_state_dict("nano")
    hot = dict(sd)
    k1, k2 = "backbone.image_model.model.stages.2.1.pwconv1", "backbone.image_model.model.stages.2.1.pwconv2.weight"
    hot[k1 + ".weight"] = sd[k1 + ".weight"] * np.float32(3e5)
    hot[k1 + ".bias"] = sd[k1 + ".bias"] * np.float32(3e5)
    hot[k2] = sd[k2] / np.float32(3e5)                     # GELU is not homogeneous, but the scaled net is still a valid net
    bank = torch.from_numpy(W.make_text_bank(20))
    rgb = W.make_images(2, 128, 128, seed=5)
    inputs = [torch.from_numpy(np.ascontiguousarray(im[..., ::-1].transpose(2, 0, 1))) for im in rgb]

    def run(state, precision, calibrate=False):
        m = YOLOWorldDetector("nano", test_cfg=dict(max_per_img=50), max_classes=20, precision=precision)
        m.load_state_dict({n: torch.from_numpy(v) for n, v in state.items()})
        m.cuda().eval()
        m.set_text_embeddings(bank)
        m._h.auto_calibrate = calibrate          # off: the run-time guard itself is under test (calibration would pre-empt it)
        out = m.test_step(dict(inputs=inputs, data_samples=[DetDataSample(), DetDataSample()]))
        return m, out
    import warnings
    with warnings.catch_warnings(record=True) as wrec:
        warnings.simplefilter("always")
        m16, o16 = run(hot, "fp16x3")
    assert any("fp16 range" in str(w.message) for w in wrec), "the fallback must be announced"
    tower = m16._h.tower(2, 128, 128)
    assert tower.precision == "fp32" and tower.overflowed and m16._h.precision == "fp32"
    m32, o32 = run(hot, "fp32")
    for a, b in zip(o16, o32):
        assert len(a.pred_instances) == len(b.pred_instances) > 0
        assert torch.equal(a.pred_instances.scores, b.pred_instances.scores) and torch.equal(a.pred_instances.labels, b.pred_instances.labels)
        assert bool(torch.isfinite(a.pred_instances.bboxes).all())
    with warnings.catch_warnings(record=True) as wrec:
        warnings.simplefilter("always")
        m_ok, _ = run(sd, "fp16x3")
    assert not any("fp16 range" in str(w.message) for w in wrec) and m_ok._h.tower(2, 128, 128).precision == "fp16x3"
    # round 3: with the first-batch calibration on (the default) the hot checkpoint's hidden tensor gets a 2^-k split scale
    # and the tower STAYS on the fp16x3 kernels — no fallback, same detections as the fp32 tower
    with warnings.catch_warnings(record=True) as wrec:
        warnings.simplefilter("always")
        m_cal, o_cal = run(hot, "fp16x3", calibrate=True)
    t_cal = m_cal._h.tower(2, 128, 128)
    assert not any("fp16 range" in str(w.message) for w in wrec) and t_cal.precision == "fp16x3" and not t_cal.overflowed
    assert any(v < 1.0 for v in t_cal.sscale.values())
    for a, b in zip(o_cal, o32):
        assert len(a.pred_instances) == len(b.pred_instances)
        assert torch.equal(a.pred_instances.labels, b.pred_instances.labels)
        assert float((a.pred_instances.scores - b.pred_instances.scores).abs().max()) < 1e-4


def test_small_batches_replay_a_hipgraph_and_match_the_eager_step():
    """batches of at most ``graph_max_batch`` images ($WEDETECT_GRAPH_MAX_BATCH; 4 in rounds 2-4, 0 = eager by
    default since round 5, when the eager step with the neck / head DAG became as fast) go through a hipGraph captured on first use.  Same buffers, same kernels: bit-identical to the eager step, on the
    first call (capture + replay) and on later ones (replay with new inputs); a larger batch stays eager."""
    from wedetect_amd import weights as W
    from wedetect_amd.detector import SimpleYOLOWorldDetector
    sd_np = W.make_state_dict("nano", num_prompts=32)
    def make():
        m = SimpleYOLOWorldDetector("nano", prompt_dim=768, num_prompts=32, num_proposals=100)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in W.to_uni_keys(sd_np).items()}, strict=False)
        return m.cuda().eval()
    graphed, eager = make(), make()
    eager._h.graph_max_batch = 0
    graphed._h.graph_max_batch = 4
    imgs = _images()
    for batch in (imgs, imgs[::-1], imgs[:1], imgs[1:2]):
        a, b = graphed(batch), eager(batch)
        assert len(a) == len(b) == len(batch)
        for x, y in zip(a, b):
            assert x.keys() == y.keys()
            for k in x:
                assert torch.equal(x[k], y[k]), f"graph replay differs from the eager step in {k}"
    assert len(graphed._h._graphs) == 2 and len(eager._h._graphs) == 0      # one graph per (tower = batch size)
    graphed._h.graph_max_batch = 2
    n = len(graphed._h._graphs)
    graphed(imgs)                                                           # 3 images > 2: eager, no new graph
    assert len(graphed._h._graphs) == n


def test_neck_input_layers_run_fp16x3_under_their_own_guard():
    """The five neck layers that read the backbone residual streams directly ran the fp32 kernel unconditionally in round 1
    (their inputs are not bounded by construction).  They now run fp16x3 with a range flag of their own: a checkpoint whose
    residual stream leaves the fp16 range trips it, ONLY those layers are pinned to fp32 (the tower stays fp16x3), the step is
    repeated and equals a tower that pinned them from the start, bit for bit."""
    import warnings
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    sd = W.make_state_dict("nano", num_prompts=0)
    hot = dict(sd)
    k = "backbone.image_model.model.stages.3.0.pwconv2"      # blow up the LAST residual stream (c4) only: it feeds reduce_layer0
    hot[k + ".weight"] = sd[k + ".weight"] * np.float32(4e6)
    hot[k + ".bias"] = sd[k + ".bias"] * np.float32(4e6)
    r = "neck.reduce_layer0.block.conv.weight"                # ... and let the layer that reads it scale it back, so that
    hot[r] = sd[r] / np.float32(4e6)                          # everything downstream sees ordinary magnitudes
    x = torch.from_numpy(W.make_images(2, 128, 128, seed=8)).cuda()
    text = torch.from_numpy(W.make_text_bank(20)).cuda()

    def tower_for(state, guard):
        import os
        old = os.environ.get("WEDETECT_NECK_GUARD")
        os.environ["WEDETECT_NECK_GUARD"] = guard
        try:
            return ImageTower("nano", pack(state, "nano"), 2, 128, 128, max_classes=20)
        finally:
            if old is None:
                del os.environ["WEDETECT_NECK_GUARD"]
            else:
                os.environ["WEDETECT_NECK_GUARD"] = old
    def step(t):
        meta = t.identity_meta()
        run = lambda: t.detect(x, text, meta, normalize_text=True, score_thr=0.001, with_embed=True)
        res = run()
        counts = t.checked_counts(res, run)
        return {k2: v.clone() for k2, v in res.items()}, counts
    # a tame checkpoint: guarded layers stay fp16x3, nothing trips, results within fp32 tolerance of the pinned tower
    tg, tp = tower_for(sd, "1"), tower_for(sd, "0")
    assert not tg.neck_pin and tp.neck_pin
    rg, cg = step(tg)
    rp, cp = step(tp)
    assert not tg.neck_pin and tg.precision == "fp16x3" and cg == cp
    assert_close("guarded vs pinned embeddings", tg.embed, to_np(tp.embed), 2e-4, 2e-4)
    # the hot checkpoint: c4 ~ 1e6
    th, tq = tower_for(hot, "1"), tower_for(hot, "0")
    with warnings.catch_warnings(record=True) as wrec:
        warnings.simplefilter("always")
        rh, ch = step(th)
    assert any("neck input layer" in str(w.message) for w in wrec)
    assert th.neck_pin and th.precision == "fp16x3" and not th.overflowed
    rq, cq = step(tq)
    assert ch == cq and tq.precision == "fp16x3"
    for k2 in rh:
        assert torch.equal(rh[k2], rq[k2]), k2
