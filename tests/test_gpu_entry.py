"""Script-level drop-in parity (GPU): BASELINE configs[0] through the ``infer_wedetect.py``-compatible entry, the
``generate_proposal.py`` / ``extract_embedding.py`` / ``retrieval_metric.py`` entries, and the device test pipeline
(cv2-style resize + letter pad) bit-exact against its oracle."""
import json
import os

import numpy as np
import pytest
import torch

from tests.test_gpu_text import _hf
from tests.util import assert_close, to_np

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def toy_tokenizer(vocab):
    def tokenizer(strings):                                # deterministic stand-in: utf-8 bytes -> ids, padded with 1
        rows = [[0] + [4 + (b * 7 + i) % (vocab - 4) for i, b in enumerate(s.encode())][:12] + [2] for s in strings]
        ln = max(len(r) for r in rows)
        ids = torch.tensor([r + [1] * (ln - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": (ids != 1).long()}
    return tokenizer


def _text_tower(seed=5, vocab=400):
    cfg, model, head = _hf(dict(vocab_size=vocab, max_position_embeddings=40, hidden_size=768, num_hidden_layers=2,
                                num_attention_heads=12, intermediate_size=512), 768, seed=seed)
    sd = {"backbone.text_model.model." + k: v.detach() for k, v in model.state_dict().items()}
    sd["backbone.text_model.head.weight"], sd["backbone.text_model.head.bias"] = head.weight.detach(), head.bias.detach()
    return model, head, sd


def _hf_bank(model, head, tok):
    with torch.no_grad():
        hs = model(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"])["last_hidden_state"][:, 0]
        return torch.nn.functional.normalize(head(hs), dim=-1)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_config0_tiny_through_the_infer_wedetect_entry(tmp_path, precision):
    """BASELINE configs[0]: WeDetect-Tiny, one 640 x 640 synthetic image, 80 class prompts + the blank class, driven
    through infer_wedetect.main (config file -> init_detector -> Compose(test_pipeline) -> reparameterize ->
    test_step -> threshold / top-k).  Checked against the oracle called directly on the same pixels and the same bank
    (HuggingFace XLM-R on the CPU): <= 300 detections before the demo filter, kept (anchor, class) lists equal,
    scores within 1e-3, boxes within 1e-2 px."""
    from PIL import Image
    import infer_wedetect as iw
    from oracle import postprocess as opp
    from oracle import ref_cpu as orc
    from wedetect_amd import weights as W
    from wedetect_amd.arch import get_arch
    vocab = 400
    hf_model, hf_head, text_sd = _text_tower(vocab=vocab)
    sd_np = W.make_state_dict("tiny")
    ckpt = {"state_dict": {**{k: torch.from_numpy(v) for k, v in sd_np.items()}, **text_sd}, "meta": {}}
    ckpt_path = str(tmp_path / "wedetect_tiny.pth")
    torch.save(ckpt, ckpt_path)
    img = W.make_images(1, 640, 640, seed=1234)[0]
    img_path = str(tmp_path / "synthetic.png")
    Image.fromarray(img).save(img_path)                                        # PNG: lossless
    names = [f"class {i} 名" for i in range(80)]
    (tmp_path / "names.txt").write_text("\n".join(names) + "\n", encoding="utf-8")
    out_dir = str(tmp_path / "out")
    tok = toy_tokenizer(vocab)
    thr, topk = 0.05, 100
    res = iw.main(["--config", os.path.join(ROOT, "config", "wedetect_tiny.py"), "--checkpoint", ckpt_path, "--image", img_path,
                   "--text", str(tmp_path / "names.txt"), "--threshold", str(thr), "--topk", str(topk), "--device", "cuda:0",
                   "--output-dir", out_dir, "--precision", precision, "--dump-json"], tokenizer=tok)
    assert len(res) == 1 and os.path.exists(os.path.join(out_dir, "synthetic.png"))
    pred = res[0]
    dumped = json.load(open(os.path.join(out_dir, "synthetic.json")))
    assert dumped["texts"] == names + [" "] and len(dumped["scores"]) == len(pred["scores"])
    # ---- the oracle, called directly
    texts = names + [" "]
    bank = _hf_bank(hf_model, hf_head, tok(texts))                              # [81, 768] unit rows
    sd = orc.to_torch(sd_np)
    with torch.no_grad():
        _, p = orc.forward_features(sd, get_arch("tiny"), img[None])
        flat = orc.head_flat(sd, p, bank, normalize_text=True)
    o = opp.mmdet_predict_image(flat["boxes"][0].numpy(), flat["scores"][0].numpy(), np.zeros(4, np.float32), (1.0, 1.0), (640, 640))
    assert o["scores"].shape[0] <= 300
    keep = o["scores"] > thr                                                    # infer_wedetect.py:119-124
    sc, bx, lb = o["scores"][keep], o["bboxes"][keep], o["labels"][keep]
    if sc.shape[0] > topk:
        idx = torch.from_numpy(sc).topk(topk)[1].numpy()
        sc, bx, lb = sc[idx], bx[idx], lb[idx]
    assert len(pred["scores"]) == sc.shape[0] > 0, (len(pred["scores"]), sc.shape[0])
    assert_close("entry scores", pred["scores"], sc, 1e-3)
    assert np.array_equal(pred["labels"], lb), "kept class list differs from the oracle"
    assert_close("entry boxes", pred["bboxes"], bx, 1e-2)
    # ---- and before the demo's filter: the full <= 300-row result, indices exact on equal inputs
    from wedetect_amd.apis import init_detector
    from wedetect_amd.detector import DetDataSample
    model = init_detector(os.path.join(ROOT, "config", "wedetect_tiny.py"), ckpt_path, device="cuda:0", tokenizer=tok, precision=precision)
    model.reparameterize([[t] for t in texts])
    assert_close("device bank vs HuggingFace", model.text_feats, bank, 2e-5)
    chw_bgr = torch.from_numpy(np.ascontiguousarray(img[..., ::-1].transpose(2, 0, 1)))
    out = model.test_step(dict(inputs=chw_bgr[None], data_samples=[DetDataSample(metainfo=dict(ori_shape=(640, 640), texts=texts))]))[0]
    pi = out.pred_instances
    assert len(pi) == o["scores"].shape[0] <= 300
    assert_close("full scores", pi.scores, o["scores"], 1e-3)
    tower = model._h.tower(1, 640, 640)
    n = len(pi)
    # the full kept list, position by position (round 5: the "> 97 % of the labels" fallback for near-tie images is gone;
    # the one allowance is the counted tie-run permutation among rows whose ORACLE scores are closer than the score
    # difference measured in this very comparison — printed and logged by compare_kept_lists, a membership change fails)
    from tests.util import compare_kept_lists
    o_eff = opp.mmdet_predict_image(flat["boxes"][0].numpy(), flat["scores"][0].numpy(), np.zeros(4, np.float32), (1.0, 1.0), (640, 640),
                                    effective=True)
    mg = o_eff["margins"]
    compare_kept_lists(f"configs[0] tiny entry [{precision}]", tower.out_anchors[0, :n], pi.labels, pi.scores, o["anchors"], o["labels"],
                       o["scores"], [mg["iou_margin"], mg["pair_gap"], mg["kept_gap"], mg["cut_gap"]], got_boxes=pi.bboxes,
                       ref_boxes=o["bboxes"], eff_margins=o_eff["eff_margins"], allow=("tie_run",))


def test_cv_resize_kernel_bit_exact_vs_oracle():
    """wd_cv_resize_paste_u8 == oracle/cv2_resize.py for every mode: exact 2x / 3x / (2, 1) box sums, the general
    float-table area path (operation order preserved: no fma contraction), 11-bit bilinear up-scaling, plain copy;
    pad value, paste offset and the BGR<->RGB swap."""
    from oracle import cv2_resize as cv
    from wedetect_amd.pipeline import cv_resize_pad
    g = np.random.default_rng(11)
    cases = [((720, 1280), (360, 640), "area"), ((960, 1920), (320, 640), "area"), ((480, 1280), (480, 640), "area"),
             ((1080, 1920), (360, 640), "area"), ((427, 640), (427, 640), "area"), ((500, 375), (640, 480), "bilinear"),
             ((1000, 750), (640, 480), "area"), ((33, 47), (20, 31), "area"), ((9, 7), (23, 31), "bilinear"),
             ((2, 2), (640, 640), "bilinear"), ((641, 700), (586, 640), "area"), ((1333, 801), (640, 384), "area")]
    for (sh, sw), (dh, dw), mode in cases:
        src = g.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        ref = cv.cv2_resize_u8(src, (dw, dh), mode) if (sh, sw) != (dh, dw) else src
        top, left = (640 - dh) // 2, (640 - dw) // 2
        got = cv_resize_pad(torch.from_numpy(src).cuda(), dh, dw, mode, (640, 640), top, left, 114)
        got = to_np(got)
        inner = got[top:top + dh, left:left + dw]
        assert np.array_equal(inner, ref), (sh, sw, dh, dw, mode, int(np.abs(inner.astype(int) - ref.astype(int)).max()),
                                            float(np.mean(inner != ref)))
        mask = np.ones((640, 640), bool)
        mask[top:top + dh, left:left + dw] = False
        assert np.all(got[mask] == 114)
        sw_ = to_np(cv_resize_pad(torch.from_numpy(src).cuda(), dh, dw, mode, (640, 640), top, left, 114, swap_rb=True))
        assert np.array_equal(sw_[top:top + dh, left:left + dw], ref[..., ::-1])
    from wedetect_amd import lib as L
    with pytest.raises(L.WedetectHipError):
        cv_resize_pad(torch.zeros(10, 10, 3, dtype=torch.uint8).cuda(), 20, 20, "bilinear", (16, 16), 0, 0)
    # chw (BGR) -> hwc (RGB) packing, uint8 and float inputs
    x = torch.from_numpy(g.integers(0, 256, (2, 3, 8, 12), dtype=np.uint8)).cuda()
    y = torch.empty(2, 8, 12, 3, dtype=torch.uint8, device="cuda")
    L.chw_to_hwc_u8(x, y)
    assert torch.equal(y, x.flip(1).permute(0, 2, 3, 1))
    L.chw_to_hwc_u8(x.float() + 0.25, y)
    assert torch.equal(y, x.flip(1).permute(0, 2, 3, 1))


def test_device_test_pipeline_matches_oracle_pixels_and_reference_geometry(tmp_path):
    """Compose(cfg.test_pipeline) on files: canvas == oracle keep-ratio resize + letter pad, metainfo == the
    reference's transform code (mmdet_test_geometry is pinned to it), texts flattened, BGR CHW packing."""
    from PIL import Image
    from oracle import cv2_resize as cv
    from wedetect_amd.cfgfile import Config
    from wedetect_amd.pipeline import Compose
    from wedetect_amd.preprocess import mmdet_test_geometry
    cfg = Config.fromfile(os.path.join(ROOT, "config", "wedetect_base.py"))
    pipe = Compose(cfg.test_pipeline)
    g = np.random.default_rng(2)
    for h, w in ((720, 1280), (480, 640), (300, 500), (1000, 750), (640, 640), (37, 90)):
        rgb = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = str(tmp_path / f"im_{h}x{w}.png")
        Image.fromarray(rgb).save(path)
        data = pipe(dict(img_id=0, img_path=path, texts=[["a"], ["b", "bb"], [" "]]))
        bgr = rgb[..., ::-1]
        ref, pad = cv.letter_pad(cv.keep_ratio_resize(bgr, (640, 640)), (640, 640), 114)
        inputs, sample = data["inputs"], data["data_samples"]
        assert tuple(inputs.shape) == (3, 640, 640) and inputs.dtype == torch.uint8 and inputs.is_cuda
        assert np.array_equal(to_np(inputs.permute(1, 2, 0)), ref), (h, w)
        geo = mmdet_test_geometry(h, w, (640, 640))
        m = sample.metainfo
        assert m["ori_shape"] == (h, w) and tuple(m["scale_factor"]) == tuple(geo["scale_factor"])
        assert np.array_equal(m["pad_param"], geo["pad_param"]) and np.array_equal(m["pad_param"], pad)
        assert m["texts"] == ["a", "b", " "] and m["img_path"] == path and tuple(m["img_shape"][:2]) == (640, 640)


def test_generate_proposal_and_retrieval_entries(tmp_path, monkeypatch):
    """generate_proposal.main, extract_embedding.main (one rank over RCCL, real batches, ragged last batch) and
    retrieval_metric.main on a miniature model: the saved file has the reference's structure, its records equal the
    detector's direct outputs, and the metric script reads it."""
    import importlib.util
    from PIL import Image
    import generate_proposal as gp
    from wedetect_amd import detector as D
    from wedetect_amd import weights as W
    monkeypatch.setattr(gp, "model_size_of", lambda p: "nano")
    monkeypatch.setitem(D._IMG_SIZE, "nano", (128, 128))
    sd = {k: torch.from_numpy(v) for k, v in W.to_uni_keys(W.make_state_dict("nano", num_prompts=256)).items()}
    uni = str(tmp_path / "wedetect_base_uni.pth")
    torch.save(sd, uni)                                                        # flat dict, reference-remapped keys
    g = np.random.default_rng(8)
    sizes = [(120, 200), (128, 64), (90, 90), (128, 128), (64, 100)]
    (tmp_path / "imgs").mkdir()
    images = []
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(tmp_path / "imgs" / f"{i}.png"))
        images.append(dict(id=100 + i, file_name=f"{i}.png"))
    out = gp.main(["--wedetect_uni_checkpoint", uni, "--image", str(tmp_path / "imgs" / "0.png"), "--score_thre", "0.05",
                   "--num_proposals", "50", "--visualize", "--output", str(tmp_path / "pred.png")])
    assert os.path.exists(tmp_path / "pred.png") and out["bboxes"].shape[1] == 4 and len(out["outputs"][0]["scores"]) <= 50
    assert bool((out["scores"] > 0.05).all()) and out["embeddings"].shape[1] == 768
    # ---- extract_embedding with a precomputed bank and with the text tower
    spec = importlib.util.spec_from_file_location("extract_embedding", os.path.join(ROOT, "eval_retrieval", "extract_embedding.py"))
    ee = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ee)
    monkeypatch.setattr(ee, "load_uni_detector", gp.load_uni_detector)
    ann = dict(images=images, categories=[dict(id=1, name="cat"), dict(id=2, name="dog"), dict(id=3, name="kite")],
               annotations=[dict(image_id=100, category_id=1), dict(image_id=101, category_id=2), dict(image_id=103, category_id=1)])
    (tmp_path / "ann.json").write_text(json.dumps(ann))
    (tmp_path / "texts.json").write_text(json.dumps([["cat", "kitty"], ["dog"], ["kite"]]))
    vocab = 400
    hf_model, hf_head, text_sd = _text_tower(vocab=vocab, seed=6)
    wck = str(tmp_path / "wedetect_base.pth")
    torch.save({"state_dict": text_sd}, wck)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    outp = str(tmp_path / "coco_uni.pth")
    tok = toy_tokenizer(vocab)
    ee.main(["--model", "uni", "--wedetect_checkpoint", wck, "--wedetect_uni_checkpoint", uni, "--dataset", "coco",
             "--batch-size", "2", "--num-workers", "0", "--ann-path", str(tmp_path / "ann.json"), "--image-path", str(tmp_path / "imgs"),
             "--class-texts", str(tmp_path / "texts.json"), "--output", outp], tokenizer=tok)
    pred = torch.load(outp, map_location="cpu")
    assert set(pred) == {"image_embedding", "text_embedding"} and len(pred["image_embedding"]) == 5
    assert_close("text bank", pred["text_embedding"], _hf_bank(hf_model, hf_head, tok(["cat", "dog", "kite"])), 2e-5)
    model = gp.load_uni_detector(uni)
    for rec, im in zip(pred["image_embedding"], images):
        direct = model([str(tmp_path / "imgs" / im["file_name"])])[0]
        assert rec["image_id"] == im["id"] and set(rec) == {"image_id", "embedding", "scale", "bias"}
        assert rec["embedding"].shape == direct["embeddings"].shape and rec["embedding"].shape[0] <= 300
        # `model` is a SECOND detector instance: it chose its fp16x3 split scales from its own first batch (image 0 alone, the
        # script's from images 0 + 1), so the two runs agree to fp32 rounding noise, not bit for bit (one instance does: an image
        # gets the same bits in any batch shape, tests/test_gpu_detector.py) — the same rows within 1e-5, and a row may change
        # places only with a row whose score is within that noise
        de, sc_d = direct["embeddings"].cpu(), direct["scores"].cpu()
        differ = ((rec["embedding"] - de).abs().amax(dim=1) > 1e-5).nonzero().flatten().tolist()
        k = 0
        while k < len(differ):                             # maximal runs of consecutive positions whose rows do not match in place
            e = k
            while e + 1 < len(differ) and differ[e + 1] == differ[e] + 1:
                e += 1
            run = differ[k:e + 1]
            assert float(sc_d[run].max() - sc_d[run].min()) < 2e-6 * len(run), f"rows {run} moved across a score gap above the noise"
            a, b_ = rec["embedding"][run], de[run]
            order_b = sorted(range(len(run)), key=lambda r: b_[r, :8].round(decimals=4).tolist())
            order_a = sorted(range(len(run)), key=lambda r: a[r, :8].round(decimals=4).tolist())
            assert_close(f"rows {run} as a set", a[order_a], b_[order_b], 1e-5)
            assert torch.equal(rec["scale"][run][order_a], direct["scales"].cpu()[run][order_b])
            assert torch.equal(rec["bias"][run][order_a], direct["bias"].cpu()[run][order_b])
            k = e + 1
        same = [i for i in range(de.shape[0]) if i not in set(differ)]
        assert torch.equal(rec["scale"][same], direct["scales"].cpu()[same]) and torch.equal(rec["bias"][same], direct["bias"].cpu()[same])
    # the reference's scoring lines run on the file (retrieval_metric.py:367-375)
    r0 = pred["image_embedding"][0]
    lg = torch.einsum("bw,kw->bk", r0["embedding"], pred["text_embedding"])
    lg = torch.sigmoid(lg * r0["scale"].exp().unsqueeze(1) + r0["bias"].unsqueeze(1)).max(dim=0)[0]
    spec = importlib.util.spec_from_file_location("retrieval_metric", os.path.join(ROOT, "eval_retrieval", "retrieval_metric.py"))
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    thre = float(lg.median())
    results = rm.main(["--model", "uni", "--dataset", "coco", "--thre", str(thre), "--ann-path", str(tmp_path / "ann.json"), "--pred", outp])
    assert set(results) == {"cat", "dog"}                                       # 'kite' has no ground truth: skipped (retrieval_metric.py:28-29)
    assert all(0.0 <= v["precision"] <= 1.0 and v["support"] >= 1 for v in results.values())


def test_eval_recall_entry(tmp_path, monkeypatch):
    """eval_recall/eval_recall.py (the reference's third Uni script, eval_recall.py:1491-1589): one rank over RCCL, batch 2
    with a ragged tail, COCO-style annotations with a crowd and an ignored box.  AR@100 / AR@300 printed by the entry ==
    the reference evaluator's numbers (oracle.evaluate, pinned to recall.py) on the detector's own proposals; a planted
    ground truth equal to one proposal of each image makes the recall non-trivial."""
    import importlib.util
    from PIL import Image
    import generate_proposal as gp
    from oracle import evaluate as oev
    from wedetect_amd import detector as D
    from wedetect_amd import weights as W
    monkeypatch.setattr(gp, "model_size_of", lambda p: "nano")
    monkeypatch.setitem(D._IMG_SIZE, "nano", (128, 128))
    sd = {k: torch.from_numpy(v) for k, v in W.to_uni_keys(W.make_state_dict("nano", num_prompts=256)).items()}
    uni = str(tmp_path / "wedetect_base_uni.pth")
    torch.save(sd, uni)
    g = np.random.default_rng(18)
    sizes = [(120, 200), (128, 64), (90, 90), (128, 128), (64, 100)]
    (tmp_path / "imgs").mkdir()
    images = []
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(tmp_path / "imgs" / f"{i}.png"))
        images.append(dict(id=500 + i, file_name=f"{i}.png"))
    model = gp.load_uni_detector(uni)
    direct = [model([str(tmp_path / "imgs" / im["file_name"])])[0]["bboxes"].cpu().numpy() for im in images]
    anns = []
    for i, (im, bx) in enumerate(zip(images, direct)):
        if i == 3:
            continue                                                           # an image without ground truth
        x1, y1, x2, y2 = [float(v) for v in bx[min(7, len(bx) - 1)]]
        anns.append(dict(image_id=im["id"], bbox=[x1, y1, x2 - x1, y2 - y1], iscrowd=0))            # exactly one proposal
        anns.append(dict(image_id=im["id"], bbox=[1.0, 2.0, 30.0, 25.0], iscrowd=0))
        anns.append(dict(image_id=im["id"], bbox=[0.0, 0.0, 50.0, 50.0], iscrowd=1))                # dropped for COCO
        anns.append(dict(image_id=im["id"], bbox=[5.0, 5.0, 20.0, 20.0], iscrowd=0, ignore=True))   # dropped
    (tmp_path / "ann.json").write_text(json.dumps(dict(images=images, annotations=anns)))
    spec = importlib.util.spec_from_file_location("eval_recall_entry", os.path.join(ROOT, "eval_recall", "eval_recall.py"))
    er = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(er)
    monkeypatch.setattr(er, "load_uni_detector", gp.load_uni_detector)
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0").items():
        monkeypatch.setenv(k, v)
    ar100, ar300 = er.main(["--wedetect_uni_checkpoint", uni, "--dataset", "coco", "--batch-size", "2", "--num-workers", "0",
                            "--ann-path", str(tmp_path / "ann.json"), "--image-path", str(tmp_path / "imgs")])
    gts = er.ground_truth_boxes(anns, [im["id"] for im in images], drop_crowd=True)
    assert [g_.shape[0] for g_ in gts] == [2, 2, 2, 0, 2]
    iou_thrs = np.linspace(.5, 0.95, 10, endpoint=True)
    ref = oev.eval_recalls(gts, [np.asarray(b, np.float32) for b in direct], [100, 300], iou_thrs)
    assert ar100 == float(sum(ref[0]) / len(ref[0])) and ar300 == float(sum(ref[1]) / len(ref[1]))
    assert 0.4 <= ar300 <= 1.0                                                 # the planted boxes are recalled at every IoU
    lv = er.ground_truth_boxes(anns, [images[0]["id"]], drop_crowd=False)      # LVIS / PACO keep crowd boxes (eval_recall.py:95-99)
    assert lv[0].shape[0] == 3


@pytest.mark.parametrize("mode", ["detect", "retrieval"])
def test_bench_collectives_on_rccl_with_one_rank(mode):
    """The N > 1 code of bench.py — communicator bound to the rank's device, the per-step region all-gathers behind the next
    batch (detect) or the region / score-block all-gathers of the class-sharded bank (retrieval), the closing all-reduces and
    barriers — on RCCL itself, as far as a one-GPU box allows: one rank that joins a real process group
    (WEDETECT_BENCH_FORCE_DIST=1).  The collective library must report itself and the one rank it saw."""
    import subprocess
    import sys
    env = dict(os.environ, WEDETECT_BENCH_FORCE_DIST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29547", WEDETECT_BENCH_TIMEOUT="240", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    if mode == "detect":
        cmd += ["--batch", "4", "--no-fp32-reference", "--no-host-fed", "--no-other-configs"]
    else:
        cmd += ["--mode", "retrieval", "--batch", "4", "--regions", "300", "--classes", "20000"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.splitlines()
    # ONE JSON line on stdout and nothing else: RCCL's banner / warnings (it prints them to stdout, from its own threads) go to stderr
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["collective_backend"] == "nccl" and out["ranks_seen"] == 1 and out["n_gpus"] == 1
    assert out["value"] > 0 and not out["config"]["fp16x3_range_guard_tripped"]
    if mode == "detect":
        assert out["per_rank"][0]["rank"] == 0 and out["per_rank"][0]["gather_handover_ms_per_step"] >= 0.0
        assert "all-gather" not in out["config"]["parallelism"]               # the line still describes a one-GPU job
