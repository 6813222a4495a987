"""Network-level parity (GPU): the HIP image tower against (a) the CPU oracle on the same
seeded weights/images and (b) the golden fixtures generated from the reference itself.

Tolerances (north_star): embeddings / scores within 1e-3 (fp32, absolute, on O(1) values);
boxes within 1e-2 px.  Index outputs: the post-process kernels are fed the tower's own
scores/boxes and must agree EXACTLY with the oracle's post-process on those same tensors;
against reference-generated goldens the kept (anchor, class) lists are compared exactly
where the golden's score gaps exceed the fp32 noise floor, and by overlap otherwise."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, check_checksum, golden, to_np

pytestmark = pytest.mark.gpu

TOL = 1e-3
PRECISION = {"value": "fp32"}


@pytest.fixture(params=["fp32", "fp16x3"], autouse=True)
def precision(request):
    """Every network-level parity test runs in both arithmetic modes at the SAME tolerances."""
    PRECISION["value"] = request.param
    yield request.param


def build(arch, b, hw, num_prompts=256, seed=2026, seed_img=1234, **kw):
    kw.setdefault("precision", PRECISION["value"])
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    sd = W.make_state_dict(arch, seed=seed, num_prompts=num_prompts)
    tower = ImageTower(arch, pack(sd, arch), b, hw, hw, **kw)
    imgs = W.make_images(b, hw, hw, seed=seed_img)
    return sd, tower, imgs


def nhwc(t, b, hw, c):
    return to_np(t).reshape(b, hw[0], hw[1], c)


def oracle_post_on(tower, scores, boxes, embed, uni=True, thr=0.0, meta=None):
    from oracle import postprocess as opp
    lvl = np.concatenate([np.full(n, l) for l, n in enumerate(tower.nl)])
    ls = np.asarray(tower.lvl_logit_scale, np.float32)
    cb = np.asarray(tower.lvl_bias, np.float32)
    out = []
    for i in range(scores.shape[0]):
        if uni:
            out.append(opp.uni_predict_image(boxes[i], embed[i], scores[i], lvl, ls, cb, num_proposals=tower.max_out,
                                             nms_pre=tower.nms_pre, score_thr=thr))
        else:
            pad, sf, ori = meta[i]
            out.append(opp.mmdet_predict_image(boxes[i], scores[i], pad, sf, ori, score_thr=thr,
                                               nms_pre=tower.nms_pre, max_per_img=tower.max_out))
    return out


@pytest.mark.parametrize("arch,b,hw", [("nano", 3, 96), ("tiny", 1, 64)])
def test_tower_vs_oracle_all_stages(arch, b, hw):
    from oracle import ref_cpu as orc
    from wedetect_amd.arch import get_arch
    sd_np, tower, imgs = build(arch, b, hw, num_prompts=64)
    a = get_arch(arch)
    sd = orc.to_torch(sd_np)
    with torch.no_grad():
        c_ref, p_ref = orc.forward_features(sd, a, imgs)
        flat = orc.head_flat(sd, p_ref, sd["embeddings"], normalize_text=False)
    tower.backbone(torch.from_numpy(imgs).cuda())
    for i in range(4):
        assert_close(f"{arch} c{i+1}", nhwc(tower.x[i], b, tower.hw[i], a.dims[i]), c_ref[i].permute(0, 2, 3, 1), TOL, TOL)
    tower.neck()
    for i, (t, r) in enumerate(zip(tower.pyramid(), p_ref)):
        assert_close(f"{arch} p{i+3}", nhwc(t, b, tower.lv[i], r.shape[1]), r.permute(0, 2, 3, 1), TOL, TOL)
    embed, boxes = tower.head()
    assert_close(f"{arch} embeddings", embed, flat["embed"], TOL, TOL)
    assert_close(f"{arch} boxes", boxes, flat["boxes"], 1e-2, 1e-5)
    logits = tower.similarity(tower.P["prompts"], normalize=False, sigmoid=False).clone()
    assert_close(f"{arch} logits", logits, flat["logits"], TOL, TOL)
    scores = tower.similarity(tower.P["prompts"], normalize=False)
    assert_close(f"{arch} scores", scores, flat["scores"], TOL, 0)
    # post-process exactness on the tower's own tensors
    res = tower.postprocess(scores, 0.0, tower.identity_meta(), nms="torchvision")
    torch.cuda.synchronize()
    ref = oracle_post_on(tower, to_np(scores), to_np(boxes), to_np(embed))
    from oracle import postprocess as opp
    for i in range(b):
        n = int(res["count"][i])
        assert n == ref[i]["scores"].shape[0]
        assert np.array_equal(to_np(res["anchors"][i, :n]), ref[i]["anchors"]), f"img{i}: kept anchors differ"
        assert np.array_equal(to_np(res["labels"][i, :n]), ref[i]["labels"])
        assert np.array_equal(to_np(res["scores"][i, :n]), ref[i]["scores"])
        assert np.array_equal(to_np(res["embeddings"][i, :n]), ref[i]["embeddings"])
        rb = opp.unletterbox(ref[i]["bboxes"], (0.0, 0.0), 1.0, (hw, hw))
        assert np.array_equal(to_np(res["bboxes"][i, :n]), rb)


def test_text_path_normalised_bank_vs_oracle():
    """mmdet path: [K,768] bank L2-normalised on device, thr 0.001, rescale-before-NMS."""
    from oracle import ref_cpu as orc
    from wedetect_amd import weights as W
    from wedetect_amd.arch import get_arch
    arch, b, hw, k = "nano", 2, 128, 81
    sd_np, tower, imgs = build(arch, b, hw, num_prompts=0, max_classes=k)
    sd = orc.to_torch(sd_np)
    text = W.make_text_bank(k) * np.float32(1.7)
    with torch.no_grad():
        _, p_ref = orc.forward_features(sd, get_arch(arch), imgs)
        flat = orc.head_flat(sd, p_ref, torch.from_numpy(text), normalize_text=True)
    embed, boxes = tower.features(torch.from_numpy(imgs).cuda())
    scores = tower.similarity(torch.from_numpy(text).cuda(), normalize=True)
    assert_close("scores (normalised text)", scores, flat["scores"], TOL, 0)
    metas = [((8.0, 8.0, 0.0, 0.0), (0.5, 0.5), (224, 256)), ((0.0, 0.0, 12.0, 12.0), (0.8, 0.8), (160, 130))]
    meta = torch.tensor([[m[0][2], m[0][0], 0, m[1][0], m[1][1], m[2][1], m[2][0], 1.0] for m in metas],
                        dtype=torch.float32).cuda()
    res = tower.postprocess(scores, 0.001, meta, with_embed=False, nms="mmcv")
    torch.cuda.synchronize()
    ref = oracle_post_on(tower, to_np(scores), to_np(boxes), None, uni=False, thr=0.001, meta=metas)
    for i in range(b):
        n = int(res["count"][i])
        assert n == ref[i]["scores"].shape[0]
        assert np.array_equal(to_np(res["anchors"][i, :n]), ref[i]["anchors"])
        assert np.array_equal(to_np(res["labels"][i, :n]), ref[i]["labels"])
        assert np.array_equal(to_np(res["bboxes"][i, :n]), ref[i]["bboxes"])


def _compare_detections(name, res, i, fx, prefix, ref_boxes=None):
    """Kept (anchor, class) lists against the reference-generated golden, logged to gpurun_out/parity_r05.jsonl.
    Exactness is ASSERTED for every golden (round 5; round 4 recorded the round-1 640 x 640 goldens without asserting them
    because their all-candidate margins sit at the noise level — they reproduce exactly all the same, so the downgrade
    only hid regressions): identical position by position, the one allowance being a permutation inside a run of
    reference scores closer than the measured noise (tests/util.py: tie_run).  A case is demoted to "recorded" only by
    name, in tests.util.KNOWN_NOISE_LEVEL_CASES, after it has actually failed within the reference's own margins."""
    from tests.util import KNOWN_NOISE_LEVEL_CASES, compare_kept_lists
    n = int(res["count"][i])
    margins = fx[f"{prefix}.margins"] if f"{prefix}.margins" in fx else None
    eff = fx[f"{prefix}.eff_margins"] if f"{prefix}.eff_margins" in fx else None
    case = f"{name} [{PRECISION['value']}]"
    return compare_kept_lists(case, res["anchors"][i, :n], res["labels"][i, :n], res["scores"][i, :n],
                              fx[f"{prefix}.anchors"], fx[f"{prefix}.labels"], fx[f"{prefix}.scores"], margins,
                              score_tol=TOL, got_boxes=res["bboxes"][i, :n] if ref_boxes is not None else None, ref_boxes=ref_boxes,
                              assert_exact=case not in KNOWN_NOISE_LEVEL_CASES, eff_margins=eff, allow=("tie_run",))


@pytest.mark.parametrize("fixture,arch,b,hw", [("net_base_b1_64.npz", "base", 1, 64), ("net_base_b2_128.npz", "base", 2, 128),
                                               ("net_base_b1_640.npz", "base", 1, 640),
                                               ("net_base_b1_640_robust_mm.npz", "base", 1, 640), ("net_base_b1_640_robust_uni.npz", "base", 1, 640),
                                               ("net_base_b1_640_robust2_mm.npz", "base", 1, 640), ("net_base_b1_640_robust2_uni.npz", "base", 1, 640)])
def test_base_against_reference_goldens(fixture, arch, b, hw):
    from oracle import postprocess as opp
    from wedetect_amd import weights as W
    fx = golden(fixture)
    sd_np, tower, imgs = build(arch, b, hw, num_prompts=int(fx["num_prompts"]), seed=int(fx["seed_w"]), seed_img=int(fx["seed_img"]))
    a = tower.a
    tower.backbone(torch.from_numpy(imgs).cuda())
    for i in range(4):
        check_checksum(f"{fixture} c{i+1}", tower.x[i], fx, f"c{i+1}", TOL, TOL)
    tower.neck()
    for i, t in enumerate(tower.pyramid()):
        check_checksum(f"{fixture} p{i+3}", t, fx, f"p{i+3}", TOL, TOL)
    embed, boxes = tower.head()
    for l in range(3):
        e = embed[:, tower.off[l]:tower.off[l] + tower.nl[l]]
        check_checksum(f"{fixture} embed{l}", e, fx, f"embed{l}", TOL, TOL)
    # Uni path end to end
    scores = tower.similarity(tower.P["prompts"], normalize=False)
    res = tower.postprocess(scores, 0.0, tower.identity_meta(), nms="torchvision")
    torch.cuda.synchronize()
    for i in range(b):
        ref_boxes = opp.unletterbox(fx[f"img{i}.bboxes"], (0.0, 0.0), 1.0, (hw, hw))
        jj, gg = _compare_detections(f"{fixture} uni img{i}", res, i, fx, f"img{i}", ref_boxes)
        assert_close(f"{fixture} uni img{i} boxes", to_np(res["bboxes"][i])[jj], ref_boxes[gg], 2e-2, 1e-5)
        assert_close(f"{fixture} uni img{i} embeddings[:, :16]", to_np(res["embeddings"][i])[jj][:, :16],
                     fx[f"img{i}.embed16"][gg], TOL, TOL)
    # mmdet path (normalised 80-class bank, thr 0.001, rescale before NMS)
    text = torch.from_numpy(W.make_text_bank(80) * np.float32(1.7)).cuda()
    scores = tower.similarity(text, normalize=True)
    meta = torch.tensor([[float(fx[f"mm.img{i}.pad"][2]), float(fx[f"mm.img{i}.pad"][0]), 0.0,
                          float(fx[f"mm.img{i}.sf"][0]), float(fx[f"mm.img{i}.sf"][1]),
                          float(fx[f"mm.img{i}.ori"][1]), float(fx[f"mm.img{i}.ori"][0]), 1.0] for i in range(b)],
                        dtype=torch.float32).cuda()
    res = tower.postprocess(scores, 0.001, meta, with_embed=False, nms="mmcv")
    torch.cuda.synchronize()
    for i in range(b):
        jj, gg = _compare_detections(f"{fixture} mmdet img{i}", res, i, fx, f"mm.img{i}", fx[f"mm.img{i}.bboxes"])
        assert_close(f"{fixture} mmdet img{i} boxes", to_np(res["bboxes"][i])[jj], fx[f"mm.img{i}.bboxes"][gg], 4e-2, 1e-5)


def test_tiny_config0_against_the_reference_plugin_classes_at_640():
    """BASELINE configs[0] (WeDetect-Tiny, one 640 x 640 image, 80 prompts + blank) against a fixture generated by the
    REFERENCE's plugin classes at that size (make_golden.case_tiny_config0: wedetect/models ConvNextVisionBackbone('tiny') +
    CSPRepBiFPANNeck(0.75, 'tiny') + the reference head module with Tiny's widths; round 4 pinned Tiny to the plugin
    classes at 64 x 64 only).  Network checksums within 1e-3; the mmdet-path kept list position by position (effective
    margins of the reference's decisions: IoU 7.8e-4, pair 2.3e-3 — three decades above the score noise)."""
    from wedetect_amd import weights as W
    fx = golden("mm_tiny_b1_640_cfg0.npz")
    k = int(fx["k_text"])
    sd_np, tower, imgs = build("tiny", 1, 640, num_prompts=0, seed=int(fx["seed_w"]), seed_img=int(fx["seed_img"]), max_classes=k)
    tower.backbone(torch.from_numpy(imgs).cuda())
    for i in range(4):
        check_checksum(f"cfg0 c{i+1}", tower.x[i], fx, f"c{i+1}", TOL, TOL)
    tower.neck()
    for i, t in enumerate(tower.pyramid()):
        check_checksum(f"cfg0 p{i+3}", t, fx, f"p{i+3}", TOL, TOL)
    embed, boxes = tower.head()
    for l in range(3):
        check_checksum(f"cfg0 embed{l}", embed[:, tower.off[l]:tower.off[l] + tower.nl[l]], fx, f"embed{l}", TOL, TOL)
    text = torch.from_numpy(W.make_text_bank(k) * np.float32(1.7)).cuda()
    scores = tower.similarity(text, normalize=True, sigmoid=False)
    for l in range(3):                                               # raw logits of every level against the reference head's
        check_checksum(f"cfg0 mm_logits{l}", scores[:, tower.off[l]:tower.off[l] + tower.nl[l]], fx, f"mm_logits{l}", TOL, TOL)
    scores = tower.similarity(text, normalize=True)
    pad, sf, ori = fx["mm.img0.pad"], fx["mm.img0.sf"], fx["mm.img0.ori"]
    meta = torch.tensor([[float(pad[2]), float(pad[0]), 0.0, float(sf[0]), float(sf[1]), float(ori[1]), float(ori[0]), 1.0]],
                        dtype=torch.float32).cuda()
    res = tower.postprocess(scores, 0.001, meta, with_embed=False, nms="mmcv")
    torch.cuda.synchronize()
    assert min(fx["mm.img0.eff_margins"][[0, 1, 3]]) > 2e-5
    jj, gg = _compare_detections("mm_tiny_b1_640_cfg0 mmdet img0", res, 0, fx, "mm.img0", fx["mm.img0.bboxes"])
    assert_close("cfg0 boxes", to_np(res["bboxes"][0])[jj], fx["mm.img0.bboxes"][gg], 4e-2, 1e-5)


def test_large_and_tiny_goldens():
    for fixture, arch in (("net_large_b1_64.npz", "large"), ("mm_tiny_b1_64.npz", "tiny")):
        fx = golden(fixture)
        sd_np, tower, imgs = build(arch, 1, 64, num_prompts=256 if arch == "large" else 0)
        tower.backbone(torch.from_numpy(imgs).cuda())
        for i in range(4):
            check_checksum(f"{fixture} c{i+1}", tower.x[i], fx, f"c{i+1}", TOL, TOL)
        tower.neck()
        for i, t in enumerate(tower.pyramid()):
            check_checksum(f"{fixture} p{i+3}", t, fx, f"p{i+3}", TOL, TOL)
        if arch == "large":
            embed, _ = tower.head()
            for l in range(3):
                check_checksum(f"{fixture} embed{l}", embed[:, tower.off[l]:tower.off[l] + tower.nl[l]], fx,
                               f"embed{l}", TOL, TOL)


def test_full_size_properties_base_640():
    """BASELINE config sizes (batch cut to 4 to keep the CPU side quick): size-independent
    properties — determinism across runs, batch-independence (image i alone == image i in
    the batch), candidate sortedness, kept-list consistency with NMS's definition."""
    from oracle import postprocess as opp
    sd_np, tower, imgs = build("base", 4, 640)
    x = torch.from_numpy(imgs).cuda()
    r1 = {k: v.clone() for k, v in tower.detect(x, tower.P["prompts"], tower.identity_meta(), normalize_text=False,
                                                score_thr=0.0, with_embed=True).items()}
    cand_s = tower.cand_score.clone()
    cand_n = tower.cand_count.clone()
    r2 = tower.detect(x, tower.P["prompts"], tower.identity_meta(), normalize_text=False, score_thr=0.0, with_embed=True)
    torch.cuda.synchronize()
    for k in r1:
        assert torch.equal(r1[k], r2[k]), f"run-to-run nondeterminism in {k}"
    for i in range(4):
        n = int(cand_n[i])
        assert n == 30000
        s = to_np(cand_s[i, :n])
        assert np.all(s[:-1] >= s[1:]), "candidates not sorted by score"
        kept = int(r1["count"][i])
        ks = to_np(r1["scores"][i, :kept])
        assert kept == 300 and np.all(ks[:-1] >= ks[1:])
        # no kept pair of the same class may overlap by more than the threshold
        lb = to_np(r1["labels"][i, :kept])
        bx = to_np(tower.boxes[i])[to_np(r1["anchors"][i, :kept])]      # NMS ran on the unclamped boxes
        for j in range(0, kept, 37):
            same = np.nonzero(lb[j + 1:] == lb[j])[0] + j + 1
            if same.size:
                assert not np.any(opp.iou_suppresses(bx[j], bx[same], 0.7 + 1e-4))
    # batch independence: the first image alone
    _, t1, _ = build("base", 1, 640)
    ra = t1.detect(x[:1].contiguous(), t1.P["prompts"], t1.identity_meta(), normalize_text=False, score_thr=0.0,
                   with_embed=True)
    torch.cuda.synchronize()
    for k in ("bboxes", "scores", "labels", "anchors", "embeddings"):
        assert torch.equal(ra[k][0], r1[k][0]), f"image 0 differs between batch 1 and batch 4 in {k}"


def test_split_k_latency_mode_matches_default_path():
    """Opt-in split-K (batch-1 latency mode): same detections as the default path up to fp32 summation-order noise."""
    if PRECISION["value"] != "fp16x3":
        pytest.skip("split-K exists for the fp16x3 kernels only")
    _, t0, imgs = build("base", 1, 320, num_prompts=64, split_k=False)
    _, t1, _ = build("base", 1, 320, num_prompts=64, split_k=True)
    assert t0.kws is None and t1.kws is not None
    x = torch.from_numpy(imgs).cuda()
    outs = []
    for t in (t0, t1):
        r = t.detect(x, t.P["prompts"], t.identity_meta(), normalize_text=False, score_thr=0.0, with_embed=True)
        torch.cuda.synchronize()
        outs.append((t.embed.clone(), t.scores.view(-1)[: t.ntot * 64].clone(), {k: v.clone() for k, v in r.items()}))
    assert_close("split-K embeddings", outs[1][0], outs[0][0], 2e-5, 1e-5)
    assert_close("split-K scores", outs[1][1], outs[0][1], 5e-6)
    n0, n1 = int(outs[0][2]["count"][0]), int(outs[1][2]["count"][0])
    a0 = set(zip(to_np(outs[0][2]["anchors"][0, :n0]).tolist(), to_np(outs[0][2]["labels"][0, :n0]).tolist()))
    a1 = set(zip(to_np(outs[1][2]["anchors"][0, :n1]).tolist(), to_np(outs[1][2]["labels"][0, :n1]).tolist()))
    assert n0 == n1 and len(a0 & a1) >= 0.97 * n0


def test_pipelined_post_process_equals_the_in_line_step():
    """detect(overlap_post=True) — top-k / NMS on the tower's second stream, beside the next call's backbone — returns, for a
    stream of DIFFERENT batches issued back to back without any host synchronisation in between, exactly the tensors of the
    in-line step (round 4).  The staging copies are taken on the post stream, where the results are produced; a pipelined
    call followed by an in-line one (which must wait for the pending post-process before it overwrites its inputs) too."""
    from wedetect_amd import weights as W
    _, t, _ = build("tiny", 3, 128, num_prompts=48)
    meta = t.identity_meta()
    batches = [torch.from_numpy(W.make_images(3, 128, 128, seed=900 + i)).cuda() for i in range(4)]
    kw = dict(normalize_text=False, score_thr=0.0, with_embed=True)
    ref = []
    for x in batches:
        r = t.detect(x, t.P["prompts"], meta, **kw)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in r.items()})
    got = []
    for x in batches:                                   # no synchronisation between the calls
        r = t.detect(x, t.P["prompts"], meta, overlap_post=True, **kw)
        with torch.cuda.stream(t.post_stream):
            got.append({k: v.clone() for k, v in r.items()})
    r = t.detect(batches[0], t.P["prompts"], meta, **kw)     # in line, right behind a pipelined call
    t.wait_post()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ref, got)):
        for k in a:
            assert torch.equal(a[k], b[k]), f"batch {i}: {k} differs between the pipelined and the in-line step"
    for k in ref[0]:
        assert torch.equal(ref[0][k], r[k]), f"in-line step behind a pipelined one: {k} differs"


@pytest.mark.parametrize("arch,b,hw,chains,dag,split_k,depth", [("base", 2, 320, 1, True, False, 1), ("tiny", 4, 128, 2, True, False, 1),
                                                                 ("tiny", 3, 128, 1, False, False, 1), ("base", 1, 320, 1, True, True, 1),
                                                                 ("base", 2, 320, 1, True, False, 2), ("base", 1, 320, 1, True, True, 2)])
def test_neck_head_pipelined_behind_the_next_backbone_equals_the_in_line_step(arch, b, hw, chains, dag, split_k, depth):
    """Round 6: in a stream of batches (detect(overlap_post=True)) the neck + head + similarity of step i run on the tower's nh
    stream beside the BACKBONE of step i + 1 (c1..c4 double-buffered), the post-process behind them on the post stream.  Six
    DIFFERENT batches issued back to back with no host synchronisation must give exactly the tensors of the in-line steps — with
    the backbone as one chain or as image chains, the neck / head as a DAG or a serial chain, in the latency split-K class
    (whose neck then takes split-K workspaces apart from the backbone's) — and so must an in-line step right behind a pipelined one."""
    from wedetect_amd import weights as W
    if split_k and PRECISION["value"] != "fp16x3":
        pytest.skip("latency split-K is an fp16x3 mode")
    _, t, _ = build(arch, b, hw, num_prompts=48, split_k=split_k)      # split_k: the latency class (its lanes own split-K workspaces)
    meta = t.identity_meta()
    batches = [torch.from_numpy(W.make_images(b, hw, hw, seed=950 + i)).cuda() for i in range(6)]
    kw = dict(normalize_text=False, score_thr=0.0, with_embed=True)
    t.bb_chains, t.pipe_neck, t.dag = "1", "0", False
    ref = []
    for x in batches:
        r = t.detect(x, t.P["prompts"], meta, **kw)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in r.items()})
    t.bb_chains, t.pipe_neck, t.dag, t.bb_depth = str(chains), "1", dag, str(depth)     # depth 2: two backbones in flight, on two streams
    assert t._pipe_neck_on() and t._n_chains() == chains and t._bb_depth() == depth
    got = []
    for x in batches:                                   # no synchronisation between the calls
        r = t.detect(x, t.P["prompts"], meta, overlap_post=True, **kw)
        with torch.cuda.stream(t.post_stream):
            got.append({k: v.clone() for k, v in r.items()})
    r = t.detect(batches[0], t.P["prompts"], meta, **kw)     # in line, right behind a pipelined call
    t.wait_post()
    torch.cuda.synchronize()
    assert t._nh_stream is not None and len(t._x_sets) == depth + 1 and (depth == 1 or t._slot1 is not None)
    for i, (a_, b_) in enumerate(zip(ref, got)):
        for k in a_:
            assert torch.equal(a_[k], b_[k]), f"batch {i}: {k} differs between the pipelined neck / head and the in-line step"
    for k in ref[0]:
        assert torch.equal(ref[0][k], r[k]), f"in-line step behind a pipelined one: {k} differs"


@pytest.mark.parametrize("arch,b,hw", [("base", 2, 320), ("tiny", 3, 128)])
def test_neck_head_dag_on_side_streams_equals_the_serial_chain(arch, b, hw):
    """Round 5: the neck / head issued as a DAG — BiFusion input branches on lanes 1 / 2 from the start of the neck, every
    BepC3's cv2 on lane 3 beside its 3 x 3 chain, head level l on the side lanes the moment P(l+3) exists, cls / reg branches
    apart — must give EXACTLY the tensors of the one-stream chain: same kernels, same arguments, only the issue order and
    the streams differ.  Checked for a stream of different batches issued back to back with no host synchronisation (also
    with the post-process pipelined on its own stream), for the separately called neck() / head(), and under hipGraph
    capture."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import GraphedDetect
    _, t, _ = build(arch, b, hw, num_prompts=48)
    meta = t.identity_meta()
    batches = [torch.from_numpy(W.make_images(b, hw, hw, seed=700 + i)).cuda() for i in range(4)]
    kw = dict(normalize_text=False, score_thr=0.0, with_embed=True)
    t.dag = False
    ref, feats = [], []
    for x in batches:
        r = t.detect(x, t.P["prompts"], meta, **kw)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in r.items()})
        feats.append((t.embed.clone(), t.boxes.clone(), [p.clone() for p in (t.p3, t.p4, t.p5)]))
    t.dag = t._dag_in_capture = True
    t.pipe_neck = "0"          # round 6: a pipelined neck / head runs as one chain on the nh stream (its own test, above); here: the lanes
    assert t._dag_on()
    for overlap in (False, True):
        got = []
        for x in batches:                               # no synchronisation between the calls
            r = t.detect(x, t.P["prompts"], meta, overlap_post=overlap, **kw)
            with torch.cuda.stream(t.post_stream if overlap else torch.cuda.current_stream()):
                got.append({k: v.clone() for k, v in r.items()})
        t.wait_post()
        torch.cuda.synchronize()
        assert len(t._side) == 3 and t._ev_i > 10           # the side lanes were really used
        for i, (a_, b_) in enumerate(zip(ref, got)):
            for k in a_:
                assert torch.equal(a_[k], b_[k]), f"batch {i} (overlap_post={overlap}): {k} differs between the DAG and the serial chain"
    # neck() and head() called apart (the tests and diagnostics do): each is internally parallel, the pair is still exact
    t.backbone(batches[2])
    t.neck()
    t.head()
    torch.cuda.synchronize()
    e_, b_, ps = feats[2]
    assert torch.equal(t.embed, e_) and torch.equal(t.boxes, b_)
    for got_p, ref_p in zip((t.p3, t.p4, t.p5), ps):
        assert torch.equal(got_p, ref_p)
    # captured: the side streams join the capture through the lane events and come back before it ends
    g = GraphedDetect(t, 48, normalize_text=False, score_thr=0.0)
    out = {k: v.clone() for k, v in g(batches[3], t.P["prompts"], meta).items()}
    torch.cuda.synchronize()
    for k in ref[3]:
        assert torch.equal(ref[3][k], out[k]), f"hipGraph replay of the DAG step: {k} differs"


@pytest.mark.parametrize("arch,b,hw,chains,order", [("base", 4, 320, 2, "free"), ("base", 4, 256, 4, "both"), ("tiny", 6, 128, 2, "dw"),
                                                    ("tiny", 4, 128, 2, "gemm")])
def test_backbone_image_chains_equal_the_single_chain(arch, b, hw, chains, order):
    """Round 6: the backbone issued as independent IMAGE chains (contiguous image groups, each running the whole ConvNeXt on its
    own stream over its rows of the same buffers; mm_backbone.py:233-255 is batch-parallel throughout) must give EXACTLY the
    tensors of the one-chain step — c1..c4, then everything downstream — for batches issued back to back with no host
    synchronisation, with the post-process pipelined, and under hipGraph capture."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import GraphedDetect
    _, t, _ = build(arch, b, hw, num_prompts=48, split_k=False)
    meta = t.identity_meta()
    batches = [torch.from_numpy(W.make_images(b, hw, hw, seed=900 + i)).cuda() for i in range(4)]
    kw = dict(normalize_text=False, score_thr=0.0, with_embed=True)
    t.bb_chains = "1"
    assert t._n_chains() == 1
    ref, cs = [], []
    for x in batches:
        r = t.detect(x, t.P["prompts"], meta, **kw)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in r.items()})
        cs.append([c.clone() for c in t.x])
    t.bb_chains, t.bb_chain_order = str(chains), order
    t.bb_chain_stages = {"free": (0, 3), "both": (2, 3), "dw": (1, 2), "gemm": (0, 1)}[order]     # the stages outside run as one chain
    t._dag_in_capture = True
    assert t._n_chains() == chains
    t.backbone(batches[1])
    torch.cuda.synchronize()
    for i, (g_, r_) in enumerate(zip(t.x, cs[1])):
        assert torch.equal(g_, r_), f"c{i + 1} differs between {chains} image chains and one"
    for overlap in (False, True):
        got = []
        for x in batches:                               # no synchronisation between the calls
            r = t.detect(x, t.P["prompts"], meta, overlap_post=overlap, **kw)
            with torch.cuda.stream(t.post_stream if overlap else torch.cuda.current_stream()):
                got.append({k: v.clone() for k, v in r.items()})
        t.wait_post()
        torch.cuda.synchronize()
        assert len(t._chain_evs) == 3 * chains
        for i, (a_, b_) in enumerate(zip(ref, got)):
            for k in a_:
                assert torch.equal(a_[k], b_[k]), f"batch {i} (overlap_post={overlap}): {k} differs between {chains} image chains and one"
    g = GraphedDetect(t, 48, normalize_text=False, score_thr=0.0)
    out = {k: v.clone() for k, v in g(batches[3], t.P["prompts"], meta).items()}
    torch.cuda.synchronize()
    for k in ref[3]:
        assert torch.equal(ref[3][k], out[k]), f"hipGraph replay of the image-chain step: {k} differs"


def test_hipgraph_replay_equals_eager_and_is_faster_at_batch1():
    """A captured step must reproduce the eager step bit for bit on new inputs, and at batch 1
    (the reference's operating point) it removes the host launch overhead."""
    import time
    from wedetect_amd import weights as W
    from wedetect_amd.engine import GraphedDetect
    sd_np, tower, imgs = build("tiny", 1, 640, num_prompts=0, max_classes=80)
    text = torch.from_numpy(W.make_text_bank(80)).cuda()
    meta = tower.identity_meta()
    meta[:, 7] = 1.0
    g = GraphedDetect(tower, 80, normalize_text=True, score_thr=0.001)
    x1 = torch.from_numpy(W.make_images(1, 640, 640, seed=9)).cuda()
    out_g = {k: v.clone() for k, v in g(x1, text, meta).items()}
    out_e = tower.detect(x1, text, meta, normalize_text=True, score_thr=0.001, with_embed=True)
    torch.cuda.synchronize()
    for k in out_g:
        assert torch.equal(out_g[k], out_e[k]), f"graph replay differs from eager in {k}"

    def timeit(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    # like for like: the graph was captured without the side-stream DAG (batch 1), so it is held against the SERIAL eager step.
    # (Round 5: the eager step WITH the neck / head DAG is faster than either — 3.9 vs 4.35 ms — which is why the detectors
    # launch eagerly by default now; printed, not asserted: it is a property of the host, not of the kernels.)
    te_dag = timeit(lambda: tower.detect(x1, text, meta, normalize_text=True, score_thr=0.001, with_embed=True))
    tower.dag = False
    te = timeit(lambda: tower.detect(x1, text, meta, normalize_text=True, score_thr=0.001, with_embed=True))
    tower.dag = True
    tg = timeit(lambda: g(x1, text, meta))
    print(f"[tiny b1 640] eager {te*1e3:.2f} ms/step, eager + DAG {te_dag*1e3:.2f} ms/step, hipGraph {tg*1e3:.2f} ms/step")
    assert tg < te * 1.05


@pytest.mark.parametrize("arch,b,hw", [("base", 2, 128), ("tiny", 3, 96), ("large", 1, 64), ("base", 1, 640)])
def test_presplit_neck_head_is_bit_identical_to_the_loader_split_path(arch, b, hw, monkeypatch):
    """Round 3: neck / head activations travel as fp16 hi/lo groups through the LDS-DMA implicit-GEMM kernel
    (split_gemm_conv.hip).  Same halves, same K order, same epilogue arithmetic as the register-staged loader-split
    kernels of rounds 1-2 ($WEDETECT_NECK_PRESPLIT=0): region embeddings, DFL boxes and scores must come out
    BIT-IDENTICAL, and P3..P5 (stored split) must be the fp32 values to the 2^-22 the format carries."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    sd = W.make_state_dict(arch, num_prompts=64)
    packed = pack(sd, arch)
    x = torch.from_numpy(W.make_images(b, hw, hw)).cuda()
    monkeypatch.setenv("WEDETECT_NECK_PRESPLIT", "0")
    t_old = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    monkeypatch.setenv("WEDETECT_NECK_PRESPLIT", "1")
    monkeypatch.setenv("WEDETECT_CONV3", "0")                     # the tap-per-stage kernel: the K order of rounds 1-3
    t_new = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    monkeypatch.setenv("WEDETECT_CONV3", "1")                     # round 4: the row-sharing 3 x 3 kernel — (kh, chunk, kw) order
    t_new3 = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    assert not t_old._neck_split() and t_new._neck_split() and t_new3._neck_split() and t_new3.conv3 and not t_new.conv3
    e0, b0 = t_old.features(x)
    s0 = t_old.similarity(t_old.P["prompts"], normalize=False).clone()
    e1, b1 = t_new.features(x)
    s1 = t_new.similarity(t_new.P["prompts"], normalize=False)
    torch.cuda.synchronize()
    assert torch.equal(e0, e1), f"{arch}: embeddings differ (max |d| {float((e0 - e1).abs().max()):.3e})"
    assert torch.equal(b0, b1) and torch.equal(s0, s1)
    for i, (p_old, p_new) in enumerate(zip(t_old.pyramid(), t_new.pyramid())):
        d = (p_old.double() - p_new.double()).abs()
        bound = 2.0 ** -21 * p_old.double().abs() + 1e-7          # hi + lo carries the fp32 value to 2^-22 relative (3e-8 absolute below the fp16 normal range)
        assert bool((d <= bound).all()), f"{arch} P{i+3}: split storage off by {float(d.max()):.3e}"
    assert int(t_new.range_flags.sum()) == 0
    # the row-sharing kernel sums the same products in another order: equal to fp32 rounding noise
    e3, b3 = t_new3.features(x)
    s3 = t_new3.similarity(t_new3.P["prompts"], normalize=False)
    torch.cuda.synchronize()
    assert float((e3 - e1).abs().max()) <= 2e-5 * float(e1.abs().max()), f"{arch}: conv3 embeddings {float((e3 - e1).abs().max()):.3e}"
    assert float((b3 - b1).abs().max()) <= 2e-3 and float((s3 - s1).abs().max()) <= 1e-5
    assert int(t_new3.range_flags.sum()) == 0


@pytest.mark.parametrize("arch,b,hw,calibrate", [("base", 2, 128, False), ("base", 1, 640, True), ("base", 3, 96, False)])
def test_fused_block_mlp_is_bit_identical_to_the_two_launch_path(arch, b, hw, calibrate, monkeypatch):
    """Round 3: the stage-1 ConvNeXt block MLP (C = 128) runs as ONE kernel that keeps the 4C hidden activation in registers
    (split_gemm_mlp.hip; $WEDETECT_FUSE_MLP=0 restores pwconv1 / pwconv2 as two launches).  Stage-1 row counts that are not
    a multiple of 128 (b = 3 at 96 x 96: 1728 rows) keep the two launches.  Residual streams, embeddings, boxes and
    scores must be BIT-IDENTICAL, with and without calibrated range scales on the LayerNorm / hidden tensors."""
    from wedetect_amd import lib as L
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    sd = W.make_state_dict(arch, num_prompts=64)
    packed = pack(sd, arch)
    x = torch.from_numpy(W.make_images(b, hw, hw)).cuda()
    monkeypatch.setenv("WEDETECT_FUSE_MLP", "0")
    t_old = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    monkeypatch.setenv("WEDETECT_FUSE_MLP", "1")
    t_new = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    assert not t_old.fuse_mlp and t_new.fuse_mlp
    assert L.mlp_fused_supported(t_new.M[0], 128, 512) == (b * (hw // 4) ** 2 % 128 == 0)
    if calibrate:
        t_old.calibrate(x)
        t_new.calibrate(x)
        assert t_new.sscale == t_old.sscale and any(v != 1.0 for v in t_new.sscale.values())
    e0, b0 = t_old.features(x)
    s0 = t_old.similarity(t_old.P["prompts"], normalize=False).clone()
    c0 = [c.clone() for c in t_old.x]
    e1, b1 = t_new.features(x)
    s1 = t_new.similarity(t_new.P["prompts"], normalize=False)
    torch.cuda.synchronize()
    for i, (u, v) in enumerate(zip(c0, t_new.x)):
        assert torch.equal(u, v), f"residual stream c{i+1} differs (max |d| {float((u - v).abs().max()):.3e})"
    assert torch.equal(e0, e1) and torch.equal(b0, b1) and torch.equal(s0, s1)
    assert int(t_new.range_flags.sum()) == 0


@pytest.mark.parametrize("arch,b,hw", [("base", 2, 128), ("tiny", 3, 96), ("large", 1, 64)])
def test_fused_stem_matches_the_three_launch_path(arch, b, hw, monkeypatch, precision):
    """Round 3: patchify + stem conv + LayerNorm as one kernel (stem.hip; $WEDETECT_FUSE_STEM=0 restores the three launches).
    The kernel computes in fp32 (v_mfma_f32_16x16x4_f32): with precision="fp32" the first residual stream — hence
    everything after it — must not change by a bit; the fp16x3 tower ran its stem GEMM on the fp16x3 kernel, which carries
    fp32 values to 2^-22, so there the two towers agree to fp32 rounding noise.  96 / 128 / 192-wide stems."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    sd = W.make_state_dict(arch, num_prompts=64)
    packed = pack(sd, arch)
    x = torch.from_numpy(W.make_images(b, hw, hw)).cuda()
    monkeypatch.setenv("WEDETECT_FUSE_STEM", "0")
    t_old = ImageTower(arch, packed, b, hw, hw, precision=precision)
    monkeypatch.setenv("WEDETECT_FUSE_STEM", "1")
    t_new = ImageTower(arch, packed, b, hw, hw, precision=precision)
    assert not t_old.fuse_stem and t_new.fuse_stem
    e0, b0 = t_old.features(x)
    c0 = t_old.x[0].clone()
    e1, b1 = t_new.features(x)
    torch.cuda.synchronize()
    if precision == "fp32":
        assert torch.equal(c0, t_new.x[0]) and torch.equal(e0, e1) and torch.equal(b0, b1)
    else:
        assert float((e0 - e1).abs().max()) < 2e-5 and float((b0 - b1).abs().max()) < 2e-4


@pytest.mark.parametrize("arch,b,hw", [("base", 2, 320), ("large", 1, 192)])
def test_layernorm_fold_tower_matches_the_layernorm_kernel_tower(arch, b, hw):
    """ImageTower with the block LayerNorms of the two-launch stages folded into pwconv1 ($WEDETECT_LN_FOLD=1) against the tower
    that runs the LayerNorm kernel ($WEDETECT_LN_FOLD=0): another rounding order in 30 layers, the same network — embeddings, boxes and
    scores within 5e-5 / 1e-3 px / 1e-5, with and without calibrated split scales; fp32 mode ignores the switch."""
    if PRECISION["value"] != "fp16x3":
        pytest.skip("the fold exists in the fp16x3 C-split epilogue")
    _, t0, imgs = build(arch, b, hw, num_prompts=48)
    _, t1, _ = build(arch, b, hw, num_prompts=48)
    t0.ln_fold, t1.ln_fold = False, True
    assert any(t1._fold_ok(i, True) for i in range(4)) and not any(t0._fold_ok(i, True) for i in range(4))
    x = torch.from_numpy(imgs).cuda()
    for calibrate in (False, True):
        if calibrate:
            t0.calibrate(x)
            t1.calibrate(x)
            assert any(k.endswith(".dw") for k in t1.sscale) or all(v == 1.0 for v in t1.sscale.values())
        outs = []
        for t in (t0, t1):
            e, bx = t.features(x)
            s = t.similarity(t.P["prompts"], normalize=False)
            torch.cuda.synchronize()
            assert not bool(t.range_flags.any())
            outs.append((e.clone(), bx.clone(), s.clone()))
        assert_close(f"fold embeddings (calibrated={calibrate})", outs[1][0], outs[0][0], 5e-5, 1e-5)
        assert_close(f"fold boxes (calibrated={calibrate})", outs[1][1], outs[0][1], 1e-3)
        assert_close(f"fold scores (calibrated={calibrate})", outs[1][2], outs[0][2], 1e-5)


@pytest.mark.parametrize("arch,b,hw,k_cls", [("nano", 3, 96, 300), ("tiny", 2, 128, 1203)])
def test_similarity_on_the_fp16x3_kernel_matches_the_fp32_similarity(arch, b, hw, k_cls):
    """Round 6: for banks of at least 256 rows the head writes the region embeddings twice (fp16 hi/lo groups + fp32, same
    per-image row map) and similarity() runs wd_similarity_split.  The fp32 twin is BIT-identical to the single-output embedding
    conv, the split copy is the split of it, scores agree with the fp32-MFMA similarity launch to 2e-6, a detect() step keeps the
    same (anchor, label) lists except where two scores tie to within that difference, and small banks keep the fp32 kernel."""
    if PRECISION["value"] != "fp16x3":
        pytest.skip("the fp16x3 similarity kernel belongs to the fp16x3 tower")
    from wedetect_amd import weights as W
    _, t0, imgs = build(arch, b, hw, num_prompts=48)
    _, t1, _ = build(arch, b, hw, num_prompts=48)
    t0.sim_split = "0"
    x = torch.from_numpy(imgs).cuda()
    text = torch.from_numpy(W.make_text_bank(k_cls)).cuda()
    t0.calibrate(x)
    t1.adopt_scales(t0.sscale)
    assert t1.want_sim_split(k_cls) and not t1.want_sim_split(80) and not t0.want_sim_split(k_cls)
    e0, b0 = t0.features(x, num_classes=k_cls)
    s0 = t0.similarity(text, normalize=True).clone()
    e1, b1 = t1.features(x, num_classes=k_cls)
    assert t1._embed_split_valid and t1.embed_s is not None and t0.embed_s is None
    s1 = t1.similarity(text, normalize=True).clone()
    torch.cuda.synchronize()
    assert not bool(t1.range_flags.any())
    assert torch.equal(e0, e1) and torch.equal(b0, b1)
    sc = t1.sscale.get("embed", 1.0)
    rows = b * t1.ntot
    assert torch.equal(ImageTowerUnsplit(t1.embed_s[:rows], sc), e1.view(rows, -1)) or float(
        (ImageTowerUnsplit(t1.embed_s[:rows], sc) - e1.view(rows, -1)).abs().max()) <= 2.0 ** -21 * float(e1.abs().max())
    assert_close("fp16x3 similarity vs fp32 similarity", s1, s0, 2e-6)
    # the same bank object again: split once (cache hit), same scores; an in-place update of the bank is seen
    assert len(t1._text_split) == 1
    s1b = t1.similarity(text, normalize=True)
    assert torch.equal(s1b, s1) and len(t1._text_split) == 1
    text.mul_(1.0)
    t1.similarity(text, normalize=True)
    assert len(t1._text_split) == 2
    # small banks: fp32 kernel, no split copy written
    t1.features(x, num_classes=80)
    assert not t1._embed_split_valid
    # whole steps
    meta = t1.identity_meta()
    meta[:, 7] = 1.0
    r0 = {k: v.clone() for k, v in t0.detect(x, text, meta, normalize_text=True, score_thr=0.001).items()}
    r1 = t1.detect(x, text, meta, normalize_text=True, score_thr=0.001)
    torch.cuda.synchronize()
    for i in range(b):
        n0, n1 = int(r0["count"][i]), int(r1["count"][i])
        a0 = set(zip(r0["anchors"][i, :n0].tolist(), r0["labels"][i, :n0].tolist()))
        a1 = set(zip(r1["anchors"][i, :n1].tolist(), r1["labels"][i, :n1].tolist()))
        assert len(a0 ^ a1) <= max(2, n0 // 50), (i, n0, n1, len(a0 ^ a1))


def ImageTowerUnsplit(t, scale):
    from wedetect_amd.engine import ImageTower
    return ImageTower.unsplit(t, scale)


def test_latency_mode_is_the_default_of_small_towers_and_batch_invariant(monkeypatch):
    """Round 6: $WEDETECT_SPLIT_K=auto (the product default; the test session pins 0, conftest.py) turns the latency-mode split-K
    on for towers of at most four 640 x 640 images and leaves large towers unsplit.  The split count of a layer comes from its
    per-image geometry, never from the batch: the same image gets the SAME BITS alone and inside a batch of three (same class),
    and results stay within fp32 summation noise of the unsplit tower."""
    if PRECISION["value"] != "fp16x3":
        pytest.skip("split-K exists for the fp16x3 kernels only")
    from wedetect_amd.engine import ImageTower
    monkeypatch.setenv("WEDETECT_SPLIT_K", "auto")
    sd, t3, imgs = build("base", 3, 320, num_prompts=64, split_k=None)
    _, t1, _ = build("base", 1, 320, num_prompts=64, split_k=None)
    _, t0, _ = build("base", 3, 320, num_prompts=64, split_k=False)
    assert t3.kws is not None and t1.kws is not None and t0.kws is None
    assert 32 * 640 * 640 > ImageTower.SPLIT_K_AUTO_PIXELS >= 4 * 640 * 640           # the benchmark batch stays unsplit
    x = torch.from_numpy(imgs).cuda()
    t3.calibrate(x)
    t1.adopt_scales(t3.sscale)
    t0.adopt_scales(t3.sscale)
    kw = dict(normalize_text=False, score_thr=0.0, with_embed=True)
    r3 = {k: v.clone() for k, v in t3.detect(x, t3.P["prompts"], t3.identity_meta(), **kw).items()}
    e3, s3 = t3.embed.clone(), t3.scores.view(-1)[: 3 * t3.ntot * 64].view(3, t3.ntot, 64).clone()
    r0 = {k: v.clone() for k, v in t0.detect(x, t0.P["prompts"], t0.identity_meta(), **kw).items()}
    e0 = t0.embed.clone()
    for i in range(3):
        r1 = t1.detect(x[i:i + 1].contiguous(), t1.P["prompts"], t1.identity_meta(), **kw)
        torch.cuda.synchronize()
        assert torch.equal(t1.embed[0], e3[i]), f"image {i}: embeddings alone != in the batch of three"
        assert torch.equal(t1.scores.view(-1)[: t1.ntot * 64].view(t1.ntot, 64), s3[i])
        for k in ("bboxes", "scores", "labels", "anchors", "count", "embeddings"):
            assert torch.equal(r1[k][0], r3[k][i]), f"image {i} {k} alone != in the batch of three"
    assert_close("latency mode vs unsplit embeddings", e3, e0, 2e-5, 1e-5)
    assert not torch.equal(e3, e0), "no layer was split: the test shapes no longer exercise the mode"
    for i in range(3):
        n0, n3 = int(r0["count"][i]), int(r3["count"][i])
        a0 = set(zip(r0["anchors"][i, :n0].tolist(), r0["labels"][i, :n0].tolist()))
        a3 = set(zip(r3["anchors"][i, :n3].tolist(), r3["labels"][i, :n3].tolist()))
        assert len(a0 & a3) >= 0.97 * max(n0, 1)


def test_mid_class_latency_split_is_batch_invariant_inside_the_class(monkeypatch):
    """Round 6: towers between four and eight 640 x 640 images' worth of pixels (the MID class) split K by the layer's geometry at
    the class's reference batch — an image gets the SAME BITS in a batch of seven and in a batch of eight, the results stay within
    fp32 summation noise of the unsplit tower, and at least one layer is split."""
    if PRECISION["value"] != "fp16x3":
        pytest.skip("split-K exists for the fp16x3 kernels only")
    from wedetect_amd.engine import ImageTower
    monkeypatch.setenv("WEDETECT_SPLIT_K", "auto")
    size = 512
    assert ImageTower.SPLIT_K_AUTO_PIXELS < 7 * size * size and 8 * size * size <= ImageTower.SPLIT_K_MID_PIXELS
    sd, t8, imgs = build("base", 8, size, num_prompts=64, split_k=None)
    _, t7, _ = build("base", 7, size, num_prompts=64, split_k=None)
    _, t0, _ = build("base", 8, size, num_prompts=64, split_k=False)
    assert t8.kws is not None and t7.kws is not None and t0.kws is None
    assert t8._split_ref == t7._split_ref == ImageTower.SPLIT_K_MID_PIXELS // (size * size) > 1
    x = torch.from_numpy(imgs).cuda()
    t8.calibrate(x)
    t7.adopt_scales(t8.sscale)
    t0.adopt_scales(t8.sscale)
    kw = dict(normalize_text=False, score_thr=0.0, with_embed=True)
    r8 = {k: v.clone() for k, v in t8.detect(x, t8.P["prompts"], t8.identity_meta(), **kw).items()}
    e8 = t8.embed.clone()
    t0.detect(x, t0.P["prompts"], t0.identity_meta(), **kw)
    e0 = t0.embed.clone()
    r7 = t7.detect(x[1:8].contiguous(), t7.P["prompts"], t7.identity_meta(), **kw)
    torch.cuda.synchronize()
    assert torch.equal(t7.embed, e8[1:8]), "embeddings in the batch of seven != in the batch of eight"
    for k in ("bboxes", "scores", "labels", "anchors", "count", "embeddings"):
        assert torch.equal(r7[k], r8[k][1:8]), f"{k}: batch of seven != batch of eight"
    assert_close("mid-class latency mode vs unsplit embeddings", e8, e0, 2e-5, 1e-5)
    assert not torch.equal(e8, e0), "no layer was split: the test shapes no longer exercise the mid class"
