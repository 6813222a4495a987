#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference; CPU).  It
  1. builds the reference's own modules (generate_proposal.py pure-torch copy, and the
     wedetect/models mmdet-plugin classes through import stubs for mmdet/mmcv/mmengine),
  2. loads the name-keyed synthetic weights of wedetect_amd.weights into them,
  3. runs them on seeded synthetic inputs,
  4. checks the repo's oracle (oracle/ref_cpu.py, oracle/postprocess.py) against those
     outputs — bit-identical on the network, index-exact on filter/top-k (reference run
     with its sort forced stable) — and aborts on any mismatch,
  5. writes small .npz fixtures (checksums, sampled elements, final detections).

Only data is written: no reference source text is copied.  torchvision is absent, so
``torchvision.ops.batched_nms`` is bound to the oracle's restatement of THAT function
(oracle.postprocess.torchvision_batched_nms: coordinate trick up to 4000 box coordinates,
per-class loop above) while the reference's head_predict runs; the mmdet-path records
(``mm.img*``) come from the oracle's restatement of mmdet's _bbox_post_process ->
mmcv.ops.batched_nms (mmdet / mmcv are absent too).  NMS parity therefore stays
"unpinned" (no binary to run against), see oracle/__init__.py.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import sys
sys.dont_write_bytecode = True

import contextlib
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import postprocess as opp          # noqa: E402
from oracle import ref_cpu as orc              # noqa: E402
from wedetect_amd import weights as W          # noqa: E402
from wedetect_amd.arch import get_arch, HD     # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


# ------------------------------------------------------------------ reference import helpers
def _nms_stub(boxes, scores, idxs, iou_threshold):
    keep = opp.torchvision_batched_nms(boxes.numpy(), scores.numpy(), idxs.numpy(), float(iou_threshold), "cpu")
    return torch.from_numpy(keep)


def import_generate_proposal():
    tv = types.ModuleType("torchvision")
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.batched_nms = _nms_stub
    tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv.ops
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_generate_proposal", os.path.join(REF, "generate_proposal.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def stable_sort():
    """Force every Tensor.sort inside the reference to be stable."""
    orig = torch.Tensor.sort

    def patched(self, *a, **k):
        k.setdefault("stable", True)
        return orig(self, *a, **k)
    torch.Tensor.sort = patched
    try:
        yield
    finally:
        torch.Tensor.sort = orig


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Generic stand-in modules for mmdet / mmcv / mmengine / timm / cv2 so that the
    reference's wedetect/models/*.py can be imported for their nn.Module code."""
    ROOTS = ("mmdet", "mmcv", "mmengine", "timm", "cv2", "mmyolo")

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


class _Registry:
    def register_module(self, *a, **k):
        if a and isinstance(a[0], type):
            return a[0]
        return lambda cls: cls

    def build(self, cfg):
        raise RuntimeError("stub registry cannot build")


import abc as _abc


class _StubMeta(_abc.ABCMeta):
    def __getattr__(cls, item):
        raise AttributeError(item)


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        if item in ("MODELS", "TASK_UTILS", "TRANSFORMS", "DATASETS", "HOOKS", "OPTIM_WRAPPER_CONSTRUCTORS",
                    "OPTIMIZERS", "DATA_SAMPLERS", "METRICS", "FUNCTIONS"):
            return _Registry()
        if item in ("BaseModule", "BaseModel"):
            return _BaseModule
        if item == "_BatchNorm":
            return torch.nn.modules.batchnorm._BatchNorm
        if item == "DropPath":
            return lambda *a, **k: torch.nn.Identity()
        if item == "trunc_normal_":
            return lambda t, *a, **k: t
        if item == "ConvModule":
            return _ConvModule
        if item == "Linear":
            return torch.nn.Linear
        if item == "Resize":
            return _ResizeBase
        if item == "autocast_box_type":
            return lambda *a, **k: (lambda fn: fn)
        if item == "cache_randomness":
            return lambda fn: fn
        if item == "imresize":
            return _imresize
        if item == "impad":
            return _impad
        if item and item[0].isupper():
            return _StubMeta(item, (torch.nn.Module,), {"__init__": lambda self, *a, **k: torch.nn.Module.__init__(self)})
        return lambda *a, **k: None


class _BaseModule(torch.nn.Module):
    """mmengine.model.BaseModule stand-in: nn.Module that swallows ``init_cfg``."""

    def __init__(self, init_cfg=None, **kw):
        super().__init__()


class _ConvModule(torch.nn.Module):
    """mmcv.cnn.ConvModule as the reference uses it: conv (bias only without a norm layer) -> BN -> activation
    (``act_cfg=None`` = none; mmcv's default is ReLU); state-dict keys conv / bn."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), **kw):
        super().__init__()
        self.conv = torch.nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=norm_cfg is None)
        self.bn = None if norm_cfg is None else torch.nn.BatchNorm2d(out_channels, momentum=norm_cfg.get("momentum", 0.1),
                                                                     eps=norm_cfg.get("eps", 1e-5))
        kind = None if act_cfg is None else act_cfg.get("type")
        self.act = {None: None, "SiLU": torch.nn.SiLU(inplace=False), "ReLU": torch.nn.ReLU(inplace=False)}[kind]

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return x if self.act is None else self.act(x)


class _ResizeBase:
    """mmdet.datasets.transforms.Resize as the two WeDetect transforms use it: keeps ``scale`` / ``keep_ratio`` /
    ``backend`` / ``interpolation`` and runs ``_resize_img`` (the subclasses override it) from ``transform``."""

    def __init__(self, scale=None, keep_ratio=False, clip_object_border=True, backend="cv2", interpolation="bilinear", **kw):
        self.scale, self.keep_ratio, self.clip_object_border = scale, keep_ratio, clip_object_border
        self.backend, self.interpolation = backend, interpolation

    def transform(self, results):
        self._resize_img(results)
        return results


def _imresize(img, size, interpolation="bilinear", backend=None, **kw):
    """mmcv.imresize stand-in for the GEOMETRY fixtures: an image of the requested (w, h) size (pixels unused)."""
    return np.zeros((size[1], size[0]) + img.shape[2:], img.dtype)


def _impad(img, padding, pad_val=0, padding_mode="constant", **kw):
    left, top, right, bottom = padding
    return np.pad(img, ((int(top), int(bottom)), (int(left), int(right)), (0, 0)), constant_values=pad_val[0] if isinstance(pad_val, tuple) else pad_val)


_STUBS_INSTALLED = False


def install_stub_finder():
    global _STUBS_INSTALLED
    if not _STUBS_INSTALLED:
        sys.meta_path.insert(0, _StubFinder())
        _STUBS_INSTALLED = True


def import_yolo_bricks():
    """wedetect/models/layers/yolo_bricks.py with mmcv / mmdet / mmengine replaced by the stand-ins above."""
    install_stub_finder()
    spec = importlib.util.spec_from_file_location("refwd_yolo_bricks", os.path.join(REF, "wedetect", "models", "layers", "yolo_bricks.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["refwd_yolo_bricks"] = m
    spec.loader.exec_module(m)
    return m


def import_wedetect_models():
    """Import the reference's mmdet-plugin model files with stubbed third parties."""
    install_stub_finder()
    root = os.path.join(REF, "wedetect", "models")

    # fake parent packages so that the files' relative imports resolve to stand-ins
    for pkg in ("refwd", "refwd.models", "refwd.models.necks", "refwd.models.backbones"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    y8 = types.ModuleType("refwd.models.necks.yolov8_pafpn")
    y8.YOLOv8PAFPN = type("YOLOv8PAFPN", (torch.nn.Module,), {})
    sys.modules["refwd.models.necks.yolov8_pafpn"] = y8

    def load(modname, rel):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(root, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        spec.loader.exec_module(m)
        return m
    bb = load("refwd.models.backbones.mm_backbone", "backbones/mm_backbone.py")
    nk = load("refwd.models.necks.yolo_world_pafpn", "necks/yolo_world_pafpn.py")
    return bb, nk


# ------------------------------------------------------------------ small utilities
def checksum(t: torch.Tensor, n_samples=64, seed=7):
    a = t.detach().contiguous().view(-1).to(torch.float64)
    g = np.random.default_rng(seed)
    idx = np.sort(g.choice(a.numel(), size=min(n_samples, a.numel()), replace=False))
    return dict(mean=np.float64(a.mean().item()), l2=np.float64(a.pow(2).sum().sqrt().item()),
                idx=idx.astype(np.int64), val=t.detach().contiguous().view(-1)[torch.from_numpy(idx)].numpy().copy())


def put(d, prefix, cs):
    for k, v in cs.items():
        d[f"{prefix}.{k}"] = v


def must_equal(name, a: torch.Tensor, b: torch.Tensor):
    if not torch.equal(a, b):
        diff = (a - b).abs().max().item()
        raise SystemExit(f"ORACLE MISMATCH at {name}: max|d|={diff:g}")
    print(f"  [bit-identical] {name} {tuple(a.shape)}")


def load_uni_model(gp, arch, num_prompts, seed):
    sd_np = W.make_state_dict(arch, seed=seed, num_prompts=num_prompts)
    model = gp.SimpleYOLOWorldDetector(backbone_size=arch, prompt_dim=768, num_prompts=num_prompts, num_proposals=300)
    uni = {k: torch.from_numpy(v) for k, v in W.to_uni_keys(sd_np).items()}
    # BatchNorm num_batches_tracked buffers are not part of the synthetic set.
    msg = model.load_state_dict(uni, strict=False)
    missing = [k for k in msg.missing_keys if not k.endswith("num_batches_tracked")]
    assert not missing and not msg.unexpected_keys, (missing, msg.unexpected_keys)
    model.eval()
    return model, sd_np


def _margins(o) -> np.ndarray:
    """Decision margins of one image's post-process (oracle/postprocess.py batched_nms stats + the nms_pre cut):
    [min |IoU - thr| over the IoU tests taken, min score gap kept-vs-suppressed, min gap between kept rows, gap at the cut].
    A second implementation with noise below these reproduces the index lists exactly."""
    m = o["margins"]
    return np.asarray([m["iou_margin"], m["pair_gap"], m["kept_gap"], m["cut_gap"]], dtype=np.float64)


# ------------------------------------------------------------------ margin-robust cases (round 4)
# B = 1 @ 640 images whose post-process decisions are far from flipping, found by a seed search with the ORACLE's network
# (bit-identical to the reference, asserted again on the chosen seed by case_network): for each path (mmdet K-class /
# Uni 256 prompts) the seed with the largest kept-row gap among those whose EFFECTIVE iou / pair / cut margins exceed
# ROBUST_MIN (oracle.postprocess.effective_margins: only decisions that can reach the first 300 output rows count — over
# all 30 000 candidates and millions of IoU tests some decision always sits inside fp32 noise, which is why the round-3
# margins were "within noise" for 29 of 33 cases although nothing ever flipped).  The kept-row gap itself cannot reach
# ROBUST_MIN: 300 sigmoid scores of one image are nearly continuous (best of 120 seeds: 8e-6, about 6 x the measured
# score noise); a swap there permutes two output rows and is the counted tie-run relaxation of tests/util.py.
ROBUST_MIN = 2e-5
# (seed_img_mm, seed_img_uni) per arch: output of `make_golden.py --search-robust ARCH K` (kept in the file so that the
# fixtures regenerate without repeating the search)
ROBUST_SEEDS = {"base": (5008, 5020), "large": (5027, 5055)}
# round 5 (VERDICT r4 #5b): the runner-up seed of every (arch, path) from the same search (`ranked ...` lines), so that the
# asserted set is 8 images, not 4 -> net_{arch}_b1_640_robust2_{mm,uni}.npz
ROBUST_SEEDS_2 = {"base": (5054, 5029), "large": (5061, 5058)}


def search_robust(arch, k_text, start=5000, trials=120):
    sd_np = W.make_state_dict(arch, seed=2026, num_prompts=256)
    sd = orc.to_torch(sd_np)
    a = get_arch(arch)
    text = torch.from_numpy(W.make_text_bank(k_text) * np.float32(1.7))
    ls = np.asarray([sd[HD + f"cls_contrasts.{l}.logit_scale"].item() for l in range(3)], dtype=np.float32)
    cb = np.asarray([sd[HD + f"cls_contrasts.{l}.bias"].item() for l in range(3)], dtype=np.float32)
    pad, sf, ori = (8.0, 8.0, 0.0, 0.0), (0.5, 0.5), (int((640 - 16) / 0.5), int(640 / 0.5))
    best = {"mm": (-1.0, None), "uni": (-1.0, None)}
    ranked = {"mm": [], "uni": []}          # every robust seed, (kept gap, seed): round 5 takes the two best per path
    for seed in range(start, start + trials):
        imgs = W.make_images(1, 640, 640, seed=seed)
        _, p = orc.forward_features(sd, a, imgs)
        fm = orc.head_flat(sd, p, text[None], normalize_text=True)
        em = opp.mmdet_predict_image(fm["boxes"][0].numpy(), fm["scores"][0].numpy(), pad, sf, ori, effective=True)["eff_margins"]
        fu = orc.head_flat(sd, p, sd["embeddings"], normalize_text=False)
        eu = opp.uni_predict_image(fu["boxes"][0].numpy(), fu["embed"][0].numpy(), fu["scores"][0].numpy(), fu["level_of"].numpy(),
                                   ls, cb, effective=True)["eff_margins"]
        for key, e in (("mm", em), ("uni", eu)):
            if min(e[0], e[1], e[3]) > ROBUST_MIN:
                ranked[key].append((float(e[2]), seed))
                if e[2] > best[key][0]:
                    best[key] = (float(e[2]), seed)
        print(f"seed {seed}: mm eff {em}  uni eff {eu}   best so far {best}", flush=True)
    for key in ranked:
        print(f"ranked {key}: {sorted(ranked[key], reverse=True)[:6]}", flush=True)
    return best


def case_robust(gp, arch, k_text):
    s_mm, s_uni = ROBUST_SEEDS[arch]
    case_network(gp, arch, 1, 640, seed_img=s_mm, k_text=k_text, out_tag=f"{arch}_b1_640_robust_mm")
    case_network(gp, arch, 1, 640, seed_img=s_uni, k_text=k_text, out_tag=f"{arch}_b1_640_robust_uni")
    s_mm, s_uni = ROBUST_SEEDS_2[arch]
    case_network(gp, arch, 1, 640, seed_img=s_mm, k_text=k_text, out_tag=f"{arch}_b1_640_robust2_mm")
    case_network(gp, arch, 1, 640, seed_img=s_uni, k_text=k_text, out_tag=f"{arch}_b1_640_robust2_uni")


# ------------------------------------------------------------------ cases
def case_network(gp, arch, b, hw, seed_w=2026, seed_img=1234, num_prompts=256, full_predict=True, k_text=80, out_tag=None):
    """Reference pure-torch copy vs oracle on (arch, b, hw)."""
    tag = out_tag or f"{arch}_b{b}_{hw}"
    print(f"== network case {tag}")
    model, sd_np = load_uni_model(gp, arch, num_prompts, seed_w)
    sd = orc.to_torch(sd_np)
    a = get_arch(arch)
    imgs = W.make_images(b, hw, hw, seed=seed_img)
    x = orc.preprocess_u8(imgs)

    fx = dict(arch=arch, b=b, hw=hw, seed_w=seed_w, seed_img=seed_img, num_prompts=num_prompts, k_text=k_text)
    c_ref = model.backbone(x)
    c_orc = orc.backbone(sd, a, x)
    for i in range(4):
        must_equal(f"{tag}.c{i+1}", c_ref[i], c_orc[i])
        put(fx, f"c{i+1}", checksum(c_ref[i].permute(0, 2, 3, 1)))       # NHWC flattening
    p_ref = model.neck(c_ref)
    p_orc = orc.neck(sd, a, c_orc)
    for i in range(3):
        must_equal(f"{tag}.p{i+3}", p_ref[i], p_orc[i])
        put(fx, f"p{i+3}", checksum(p_ref[i].permute(0, 2, 3, 1)))

    prompts = sd["embeddings"]
    for l in range(3):
        e_ref, bb_ref, lg_ref = model.head_module_forward_single(
            p_ref[l], model.bbox_head.cls_preds[l], model.bbox_head.reg_preds[l], model.bbox_head.cls_contrasts[l])
        e_orc, lg_orc, bb_orc = orc.head_level(sd, l, p_orc[l], prompts, normalize_text=False)
        must_equal(f"{tag}.embed{l}", e_ref, e_orc)
        must_equal(f"{tag}.logits{l}", lg_ref, lg_orc)
        must_equal(f"{tag}.bbox{l}", bb_ref, bb_orc)
        put(fx, f"embed{l}", checksum(e_ref.permute(0, 2, 3, 1)))
        put(fx, f"logits{l}", checksum(lg_ref.permute(0, 2, 3, 1)))
        put(fx, f"bbox{l}", checksum(bb_ref.permute(0, 2, 3, 1)))

    # mmdet-path head: YOLOWorldHeadModule.forward with a [B,K,768] text tensor that is
    # L2-normalised inside BNContrastiveHead (generate_proposal.py:605-623, 716-752 —
    # the same code as yolo_world_head.py:90-108, 263-294).
    text = torch.from_numpy(W.make_text_bank(k_text) * np.float32(1.7))     # not unit-norm on purpose
    text_b = text[None].repeat(b, 1, 1)
    outs = model.bbox_head(p_ref, text_b)
    for l in range(3):
        _, lg_orc, bb_orc = orc.head_level(sd, l, p_orc[l], text_b, normalize_text=True)
        must_equal(f"{tag}.mm_logits{l}", outs[l][0], lg_orc)
        must_equal(f"{tag}.mm_bbox{l}", outs[l][1], bb_orc)
        put(fx, f"mm_logits{l}", checksum(outs[l][0].permute(0, 2, 3, 1)))
    if full_predict:
        flat_mm = orc.head_flat(sd, p_orc, text_b, normalize_text=True)
        for i in range(b):
            # synthetic letterbox metadata: pad (top,bottom,left,right), scale (w,h), ori (h,w)
            pad = (8.0, 8.0, 0.0, 0.0) if i % 2 == 0 else (0.0, 0.0, 12.0, 12.0)
            sf = (0.5, 0.5) if i % 2 == 0 else (0.8, 0.8)
            ori = (int((hw - 16) / 0.5), int(hw / 0.5)) if i % 2 == 0 else (int(hw / 0.8), int((hw - 24) / 0.8))
            o = opp.mmdet_predict_image(flat_mm["boxes"][i].numpy(), flat_mm["scores"][i].numpy(), pad, sf, ori, effective=out_tag is not None)
            if out_tag is not None:
                fx[f"mm.img{i}.eff_margins"] = o["eff_margins"]
                print(f"  mm img{i}: EFFECTIVE margins (iou, pair, kept, cut) {o['eff_margins']}")
            fx[f"mm.img{i}.pad"], fx[f"mm.img{i}.sf"], fx[f"mm.img{i}.ori"] = np.asarray(pad), np.asarray(sf), np.asarray(ori)
            for key in ("bboxes", "scores", "labels", "anchors"):
                fx[f"mm.img{i}.{key}"] = o[key]
            fx[f"mm.img{i}.margins"] = _margins(o)
            print(f"  mm img{i}: kept {o['scores'].shape[0]}, margins (iou, pair, kept, cut) {fx[f'mm.img{i}.margins']}")
    if full_predict:
        with stable_sort():
            res = model.head_predict(p_ref)
        flat = orc.head_flat(sd, p_orc, prompts, normalize_text=False)
        ls = np.asarray([sd[HD + f"cls_contrasts.{l}.logit_scale"].item() for l in range(3)], dtype=np.float32)
        cb = np.asarray([sd[HD + f"cls_contrasts.{l}.bias"].item() for l in range(3)], dtype=np.float32)
        for i in range(b):
            o = opp.uni_predict_image(flat["boxes"][i].numpy(), flat["embed"][i].numpy(), flat["scores"][i].numpy(),
                                      flat["level_of"].numpy(), ls, cb, effective=out_tag is not None)
            if out_tag is not None:
                fx[f"img{i}.eff_margins"] = o["eff_margins"]
                print(f"  img{i}: EFFECTIVE margins (iou, pair, kept, cut) {o['eff_margins']}")
            must_equal(f"{tag}.img{i}.bboxes", res[i]["bboxes"], torch.from_numpy(o["bboxes"]))
            must_equal(f"{tag}.img{i}.scores", res[i]["scores"], torch.from_numpy(o["scores"]))
            must_equal(f"{tag}.img{i}.embeddings", res[i]["embeddings"], torch.from_numpy(o["embeddings"]))
            n = o["scores"].shape[0]
            gaps = np.diff(o["scores"].astype(np.float64))
            fx[f"img{i}.bboxes"] = o["bboxes"]
            fx[f"img{i}.scores"] = o["scores"]
            fx[f"img{i}.labels"] = o["labels"]
            fx[f"img{i}.anchors"] = o["anchors"]
            fx[f"img{i}.embed16"] = o["embeddings"][:, :16].copy()
            fx[f"img{i}.embed_l2"] = np.linalg.norm(o["embeddings"].astype(np.float64), axis=1)
            fx[f"img{i}.num_candidates"] = o["num_candidates"]
            fx[f"img{i}.min_score_gap"] = np.float64(np.min(-gaps)) if n > 1 else np.float64(1.0)
            fx[f"img{i}.margins"] = _margins(o)
            print(f"  img{i}: margins (iou, pair, kept, cut) {fx[f'img{i}.margins']}")
            print(f"  img{i}: kept {n}, candidates {int(o['num_candidates'])}, min score gap {fx[f'img{i}.min_score_gap']:.3g}")
    np.savez_compressed(os.path.join(OUT, f"net_{tag}.npz"), **fx)


def case_filter_topk(gp):
    """filter_scores_and_topk: reference (stable sort) vs oracle, incl. ties, empty, truncation."""
    print("== filter_scores_and_topk cases")
    g = np.random.default_rng(99)
    fx = {}
    cases = {
        "ties": (np.round(g.random((500, 7), dtype=np.float32) * 20) / 20, 0.3, 1000),
        "trunc": (g.random((4000, 9), dtype=np.float32), 0.05, 30000),
        "empty": (g.random((50, 3), dtype=np.float32) * 0.001, 0.5, 100),
        "all_equal": (np.full((64, 4), 0.5, dtype=np.float32), 0.0, 100),
    }
    for name, (sc, thr, topk) in cases.items():
        with stable_sort():
            s, lab, keep, _ = gp.filter_scores_and_topk(torch.from_numpy(sc), thr, topk)
        os_, ol, oa = opp.filter_scores_and_topk(sc, thr, topk)
        must_equal(f"topk.{name}.scores", s, torch.from_numpy(os_))
        must_equal(f"topk.{name}.labels", lab, torch.from_numpy(ol))
        must_equal(f"topk.{name}.anchors", keep, torch.from_numpy(oa))
        fx[f"{name}.in"] = sc
        fx[f"{name}.thr"] = np.float32(thr)
        fx[f"{name}.topk"] = np.int64(topk)
        fx[f"{name}.scores"] = os_
        fx[f"{name}.labels"] = ol
        fx[f"{name}.anchors"] = oa
    np.savez_compressed(os.path.join(OUT, "filter_topk.npz"), **fx)


def nms_hand_cases():
    """Hand-derived vectors for the two library forms (expected keeps worked out on paper, see the comments)."""
    f = np.float32
    cases = {}
    # (1) offset quantisation: label 1202, max coordinate 1280 -> S = 1281, offset 1202 * 1281 = 1539762 (exact), fp32
    # spacing there 0.125 px.  y2 = 170.05 becomes 170.0 on the offset box: IoU = 7000 / 10000 = fp32(0.7), not > 0.7 ->
    # kept; on the original boxes IoU = 7005 / 10000 > 0.7 -> suppressed.
    cases["quant"] = dict(boxes=np.array([[100, 100, 200, 200], [100, 100, 200, 170.05], [1270, 1270, 1280, 1280]], f),
                          scores=np.array([0.9, 0.8, 0.7], f), labels=np.array([1202, 1202, 0], np.int64),
                          vanilla=[0, 2], trick=[0, 1, 2])
    # (2) cross-class meeting: max 99 -> S = 100; the class-1 box at (-50..-1) lands exactly on the class-0 box
    # (50..99) after its offset of 100 -> IoU 1 in the agnostic pass; a per-class loop keeps both.
    cases["cross"] = dict(boxes=np.array([[50, 50, 99, 99], [-50, -50, -1, -1]], f), scores=np.array([0.9, 0.8], f),
                          labels=np.array([0, 1], np.int64), vanilla=[0, 1], trick=[0])
    # (3) threshold type: inter 3, union 10 -> ovr = fp32(0.3) = 0.300000012.  torchvision compares with the double 0.3
    # (suppressed), mmcv with float(0.3) = the same fp32 value (not >: kept).
    cases["thr03"] = dict(boxes=np.array([[0, 0, 1, 6.5], [0, 3.5, 1, 10]], f), scores=np.array([0.9, 0.8], f),
                          labels=np.array([0, 0], np.int64), thr=0.3, torchvision=[0], mmcv=[0, 1])
    return cases


def case_nms():
    """NMS unit cases: hand-derived vectors + oracle-authored regression records for the three forms (parity unpinned)."""
    print("== NMS unit cases (oracle restatement of torchvision / mmcv; hand vectors)")
    f = np.float32
    fx = {}
    for name, c in nms_hand_cases().items():
        thr = c.get("thr", 0.7)
        bx, sc, lb = c["boxes"], c["scores"], c["labels"]
        tv = opp.torchvision_batched_nms(bx, sc, lb, thr, "cpu")                       # numel <= 4000: coordinate trick
        mm = opp.mmcv_batched_nms(bx, sc, lb, dict(type="nms", iou_threshold=thr))     # n < 10000: one agnostic call
        mm_split = opp.mmcv_batched_nms(bx, sc, lb, dict(type="nms", iou_threshold=thr, split_thr=1))
        va = opp.batched_nms(bx, sc, lb, thr)
        if "trick" in c:
            assert tv.tolist() == c["trick"] and mm.tolist() == c["trick"] and va.tolist() == c["vanilla"], (name, tv, mm, va)
        else:
            assert tv.tolist() == c["torchvision"] and mm.tolist() == c["mmcv"], (name, tv, mm)
        fx[f"hand.{name}.boxes"], fx[f"hand.{name}.scores"], fx[f"hand.{name}.labels"] = bx, sc, lb
        fx[f"hand.{name}.thr"] = np.float64(thr)
        fx[f"hand.{name}.tv"], fx[f"hand.{name}.mmcv"], fx[f"hand.{name}.mmcv_split"], fx[f"hand.{name}.vanilla"] = tv, mm, mm_split, va
        print(f"  hand {name}: torchvision {tv.tolist()} mmcv {mm.tolist()} mmcv/per-class {mm_split.tolist()} vanilla {va.tolist()}")
    # IoU of [0,0,10,10] vs [0,0,10,7] = 0.7 exactly (not > 0.7 -> both kept);
    # vs [0,0,10,7.0001] slightly above -> suppressed.
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 7], [0, 0, 10, 7.001], [0, 0, 10, 10], [20, 20, 30, 30],
                      [20, 20, 30, 30], [5, 5, 5, 5], [5, 5, 5, 5]], dtype=f)
    scores = np.array([0.9, 0.8, 0.8, 0.7, 0.6, 0.6, 0.5, 0.5], dtype=f)
    labels = np.array([0, 0, 0, 1, 2, 2, 3, 3], dtype=np.int64)
    keep = opp.batched_nms(boxes, scores, labels, 0.7)
    fx["unit.boxes"], fx["unit.scores"], fx["unit.labels"], fx["unit.keep"] = boxes, scores, labels, keep
    fx["unit.keep_tv"] = opp.torchvision_batched_nms(boxes, scores, labels, 0.7)
    fx["unit.keep_mmcv"] = opp.mmcv_batched_nms(boxes, scores, labels, dict(type="nms", iou_threshold=0.7))
    print("  unit keep:", keep.tolist(), fx["unit.keep_tv"].tolist(), fx["unit.keep_mmcv"].tolist())
    g = np.random.default_rng(5)
    n = 3000
    ctr = g.random((n, 2), dtype=np.float32) * 200
    wh = g.random((n, 2), dtype=np.float32) * 60 + 4
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(f)
    scores = np.sort(np.round(g.random(n, dtype=np.float32) * 500) / 500)[::-1].copy()
    labels = g.integers(0, 5, size=n).astype(np.int64)
    keep = opp.batched_nms(boxes, scores, labels, 0.7)
    keep300 = opp.batched_nms(boxes, scores, labels, 0.7, max_keep=300)
    assert np.array_equal(keep[:300], keep300)
    assert np.array_equal(keep, opp.torchvision_batched_nms(boxes, scores, labels, 0.7))      # 12000 coordinates: vanilla branch
    fx["rand.boxes"], fx["rand.scores"], fx["rand.labels"], fx["rand.keep"] = boxes, scores, labels, keep
    fx["rand.keep_mmcv"] = opp.mmcv_batched_nms(boxes, scores, labels, dict(type="nms", iou_threshold=0.7))
    fx["rand.keep_mmcv_split"] = opp.mmcv_batched_nms(boxes, scores, labels, dict(type="nms", iou_threshold=0.7, split_thr=100))
    fx["rand.keep_tv_trick"] = opp.torchvision_batched_nms(boxes, scores, labels, 0.7, "cuda")    # 12000 <= 20000: trick
    fx["empty.keep"] = opp.batched_nms(np.zeros((0, 4), f), np.zeros((0,), f), np.zeros((0,), np.int64), 0.7)
    # LVIS-sized labels on 1280-px coordinates, dense near-threshold pairs: the three forms must differ here
    g = np.random.default_rng(1203)
    n = 960
    base = g.random((n // 2, 2), dtype=np.float32) * 1100 + 20
    wh = g.random((n // 2, 2), dtype=np.float32) * 120 + 30
    a = np.concatenate([base, base + wh], 1)
    d = wh[:, 0] * f(0.3 / 1.7) + (g.random(n // 2, dtype=np.float32) - f(0.5)) * f(0.12)       # IoU ~ 0.7 +- a few 1e-4
    b = a.copy(); b[:, 0] += d; b[:, 2] += d
    boxes = np.concatenate([a, b], 0).astype(f)
    boxes[0] = [1200, 1200, 1280, 1280]
    labels = np.concatenate([g.integers(1100, 1203, n // 2)] * 2).astype(np.int64)
    scores = np.sort(g.random(n, dtype=np.float32))[::-1].copy()
    perm = np.concatenate([np.arange(n // 2) * 2, np.arange(n // 2) * 2 + 1])                   # a_i at 2i, b_i at 2i+1
    bx2, lb2 = np.empty_like(boxes), np.empty_like(labels)
    bx2[perm], lb2[perm] = boxes, labels
    kv = opp.batched_nms(bx2, scores, lb2, 0.7)
    kt = opp.torchvision_batched_nms(bx2, scores, lb2, 0.7)                                     # 3840 coordinates: trick
    km = opp.mmcv_batched_nms(bx2, scores, lb2, dict(type="nms", iou_threshold=0.7))
    assert np.array_equal(kt, km) and not np.array_equal(kv, kt), "the LVIS-label case must separate the offset forms from the label test"
    fx["lvis.boxes"], fx["lvis.scores"], fx["lvis.labels"] = bx2, scores, lb2
    fx["lvis.keep"], fx["lvis.keep_tv"], fx["lvis.keep_mmcv"] = kv, kt, km
    print(f"  lvis: vanilla keeps {kv.shape[0]}, offset forms keep {kt.shape[0]} ({len(set(kv.tolist()) ^ set(kt.tolist()))} differ)")
    print(f"  rand: {n} boxes -> {keep.shape[0]} kept")
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **fx)


def case_retrieval():
    """retrieval_metric.py:369-375 (module not importable: argparse + file IO at import);
    its five tensor lines are executed here verbatim-in-meaning with torch and compared."""
    print("== retrieval similarity cases")
    fx = {}
    g = np.random.default_rng(11)
    for k in (80, 81, 256, 1203):
        e = W.make_regions(300, seed=11 + k)          # regenerated from the seed by the tests
        t = W.make_text_bank(k)
        scale = g.choice(np.array([-0.35, -0.55, -0.2], dtype=np.float32), size=300)
        bias = g.choice(np.array([-2.6, -2.2, -1.9], dtype=np.float32), size=300)
        et, tt = torch.from_numpy(e), torch.from_numpy(t)
        cls_logits = torch.einsum('bw,kw->bk', et, tt)
        cls_logits = torch.sigmoid(cls_logits * torch.from_numpy(scale).exp().unsqueeze(1)
                                   + torch.from_numpy(bias).unsqueeze(1))
        ref = torch.max(cls_logits, dim=0)[0]
        must_equal(f"retrieval.k{k}", ref, torch.from_numpy(opp.retrieval_scores(e, t, scale, bias)))
        fx[f"k{k}.seed_embed"] = np.int64(11 + k)
        fx[f"k{k}.embed_head"] = e[:4, :8].copy()
        fx[f"k{k}.scale"], fx[f"k{k}.bias"] = scale, bias
        fx[f"k{k}.max_scores"] = ref.numpy()
    np.savez_compressed(os.path.join(OUT, "retrieval.npz"), **fx)


def case_mmdet_modules(arch="tiny", hw=64):
    """wedetect/models plugin classes (Tiny exists only there): backbone + neck vs oracle."""
    print(f"== mmdet-plugin modules, {arch} @ {hw}")
    bbm, nkm = import_wedetect_models()
    a = get_arch(arch)
    sd_np = W.make_state_dict(arch, seed=2026)
    sd = orc.to_torch(sd_np)
    vis = bbm.ConvNextVisionBackbone(model_name=arch)
    vis_sd = {k[len("backbone.image_model."):]: torch.from_numpy(v) for k, v in sd_np.items()
              if k.startswith("backbone.image_model.")}
    msg = vis.load_state_dict(vis_sd, strict=False)
    assert not msg.unexpected_keys, msg.unexpected_keys
    assert all(k.startswith(("model.norm.", "model.head.")) for k in msg.missing_keys), msg.missing_keys
    neck = nkm.CSPRepBiFPANNeck(scale_factor=a.neck_scale, model_size=arch)
    nk_sd = {k[len("neck."):]: torch.from_numpy(v) for k, v in sd_np.items() if k.startswith("neck.")}
    msg = neck.load_state_dict(nk_sd, strict=False)
    assert not msg.unexpected_keys and all(k.endswith("num_batches_tracked") for k in msg.missing_keys), msg
    torch.nn.Module.eval(vis)
    torch.nn.Module.eval(neck)
    for m in list(vis.modules()) + list(neck.modules()):
        m.training = False
    imgs = W.make_images(1, hw, hw, seed=1234)
    x = orc.preprocess_u8(imgs)
    c_ref = vis(x)
    c_orc = orc.backbone(sd, a, x)
    fx = dict(arch=arch, b=1, hw=hw, seed_w=2026, seed_img=1234)
    for i in range(4):
        must_equal(f"mm.{arch}.c{i+1}", c_ref[i], c_orc[i])
        put(fx, f"c{i+1}", checksum(c_ref[i].permute(0, 2, 3, 1)))
    p_ref = neck(c_ref)
    p_orc = orc.neck(sd, a, c_orc)
    for i in range(3):
        must_equal(f"mm.{arch}.p{i+3}", p_ref[i], p_orc[i])
        put(fx, f"p{i+3}", checksum(p_ref[i].permute(0, 2, 3, 1)))
    np.savez_compressed(os.path.join(OUT, f"mm_{arch}_b1_{hw}.npz"), **fx)


def case_tiny_config0(gp, hw=640, k_text=81, seed_img=1234):
    """BASELINE configs[0] pinned to the REFERENCE (round 5; VERDICT r4 missing #5): WeDetect-Tiny, one hw x hw image, through
    the plugin classes the shipped config builds (wedetect/models ConvNextVisionBackbone('tiny') + CSPRepBiFPANNeck(0.75,
    'tiny'): wedetect_tiny.py:108, yolo_world_pafpn.py:987-1137) and the reference's head module with Tiny's widths
    (generate_proposal.YOLOWorldHeadModule(768, [96, 192, 384], use_bn_head=True) — the same code as
    yolo_world_head.py:174-294), text L2-normalised inside BNContrastiveHead.  Every tensor must be torch.equal to the
    oracle's; the post-process (mmdet path, thr 0.001, mmcv NMS) is the oracle's, pinned elsewhere (filter_topk / nms /
    head_predict goldens).  Written with the margins of its decisions like the robust 640 x 640 fixtures."""
    print(f"== Tiny config[0] through the plugin classes @ {hw}")
    arch = "tiny"
    bbm, nkm = import_wedetect_models()
    a = get_arch(arch)
    sd_np = W.make_state_dict(arch, seed=2026)
    sd = orc.to_torch(sd_np)
    vis = bbm.ConvNextVisionBackbone(model_name=arch)
    vis_sd = {k[len("backbone.image_model."):]: torch.from_numpy(v) for k, v in sd_np.items()
              if k.startswith("backbone.image_model.")}
    msg = vis.load_state_dict(vis_sd, strict=False)
    assert not msg.unexpected_keys and all(k.startswith(("model.norm.", "model.head.")) for k in msg.missing_keys), msg
    neck = nkm.CSPRepBiFPANNeck(scale_factor=a.neck_scale, model_size=arch)
    nk_sd = {k[len("neck."):]: torch.from_numpy(v) for k, v in sd_np.items() if k.startswith("neck.")}
    msg = neck.load_state_dict(nk_sd, strict=False)
    assert not msg.unexpected_keys and all(k.endswith("num_batches_tracked") for k in msg.missing_keys), msg
    head = gp.YOLOWorldHeadModule(embed_dims=768, in_channels=list(a.head_in), use_bn_head=True)
    hd_sd = {k[len("bbox_head."):]: torch.from_numpy(v) for k, v in W.to_uni_keys(sd_np).items() if k.startswith("bbox_head.")}
    msg = head.load_state_dict(hd_sd, strict=False)
    assert not msg.unexpected_keys and all(k.endswith("num_batches_tracked") for k in msg.missing_keys), msg
    for mod in (vis, neck, head):
        torch.nn.Module.eval(mod)
        for m in mod.modules():
            m.training = False
    imgs = W.make_images(1, hw, hw, seed=seed_img)
    x = orc.preprocess_u8(imgs)
    fx = dict(arch=arch, b=1, hw=hw, seed_w=2026, seed_img=seed_img, k_text=k_text, num_prompts=0)
    with torch.no_grad():
        c_ref = vis(x)
        c_orc = orc.backbone(sd, a, x)
        for i in range(4):
            must_equal(f"cfg0.{arch}.c{i+1}", c_ref[i], c_orc[i])
            put(fx, f"c{i+1}", checksum(c_ref[i].permute(0, 2, 3, 1)))
        p_ref = neck(c_ref)
        p_orc = orc.neck(sd, a, c_orc)
        for i in range(3):
            must_equal(f"cfg0.{arch}.p{i+3}", p_ref[i], p_orc[i])
            put(fx, f"p{i+3}", checksum(p_ref[i].permute(0, 2, 3, 1)))
        text = torch.from_numpy(W.make_text_bank(k_text) * np.float32(1.7))
        text_b = text[None]
        outs = head(p_ref, text_b)
        for l in range(3):
            e_orc, lg_orc, bb_orc = orc.head_level(sd, l, p_orc[l], text_b, normalize_text=True)
            must_equal(f"cfg0.mm_logits{l}", outs[l][0], lg_orc)
            must_equal(f"cfg0.mm_bbox{l}", outs[l][1], bb_orc)
            put(fx, f"mm_logits{l}", checksum(outs[l][0].permute(0, 2, 3, 1)))
            put(fx, f"embed{l}", checksum(e_orc.permute(0, 2, 3, 1)))
        flat = orc.head_flat(sd, p_orc, text_b, normalize_text=True)
    pad, sf, ori = (0.0, 0.0, 0.0, 0.0), (1.0, 1.0), (hw, hw)               # the demo entry feeds a network-sized image
    o = opp.mmdet_predict_image(flat["boxes"][0].numpy(), flat["scores"][0].numpy(), pad, sf, ori, effective=True)
    fx["mm.img0.pad"], fx["mm.img0.sf"], fx["mm.img0.ori"] = np.asarray(pad), np.asarray(sf), np.asarray(ori)
    for key in ("bboxes", "scores", "labels", "anchors"):
        fx[f"mm.img0.{key}"] = o[key]
    fx["mm.img0.margins"] = _margins(o)
    fx["mm.img0.eff_margins"] = o["eff_margins"]
    print(f"  kept {o['scores'].shape[0]}, margins {fx['mm.img0.margins']}, effective {o['eff_margins']}")
    np.savez_compressed(os.path.join(OUT, f"mm_{arch}_b1_{hw}_cfg0.npz"), **fx)


def case_letterbox(gp):
    """The reference's own ``letterbox`` (generate_proposal.py:17-82, PIL BILINEAR resize + paste) on
    seeded images: down-scale, up-scale, tall, wide, already-square."""
    from PIL import Image
    g = np.random.default_rng(99)
    fx = {}
    for i, ((h, w), shape) in enumerate((((150, 200), (96, 96)), ((37, 53), (96, 96)), ((300, 90), (64, 96)),
                                         ((96, 96), (96, 96)), ((31, 257), (64, 64)))):
        # smooth + noisy content so that antialiasing weights matter
        yy, xx = np.mgrid[0:h, 0:w]
        base = (127 + 100 * np.sin(xx / 7.0 + i) * np.cos(yy / 5.0))[..., None] + g.normal(0, 25, (h, w, 3))
        img = np.clip(base, 0, 255).astype(np.uint8)
        out, ratio, (dw, dh) = gp.letterbox(Image.fromarray(img), shape)
        fx[f"img{i}"] = img
        fx[f"out{i}"] = np.asarray(out)
        fx[f"meta{i}"] = np.asarray([shape[0], shape[1], ratio, dw, dh], np.float64)
    fx["count"] = np.asarray(5)
    np.savez_compressed(os.path.join(OUT, "letterbox.npz"), **fx)


def import_recall():
    """eval_recall/recall.py as a module; its only non-numpy import is the table pretty-printer."""
    import importlib.util
    import types
    sys.modules.setdefault("terminaltables", types.SimpleNamespace(AsciiTable=lambda *a, **k: types.SimpleNamespace(table="", inner_footing_row_border=False)))
    spec = importlib.util.spec_from_file_location("ref_recall", os.path.join(REF, "eval_recall", "recall.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.print_recall_summary = lambda *a, **k: None
    return mod


def case_recall():
    """Seeded ground truths / scored proposals (some images empty, some with more gts than proposals, exact
    duplicates for IoU ties) through the reference's eval_recalls; the unsorted matched IoUs are recorded too."""
    rc = import_recall()
    g = np.random.default_rng(123)
    gts, props = [], []
    for i in range(9):
        ng = int(g.integers(0, 9)) if i != 4 else 12
        npr = int(g.integers(0, 40)) if i not in (4, 6) else (5 if i == 4 else 0)
        gt = g.uniform(0, 300, (ng, 2)); gt = np.concatenate([gt, gt + g.uniform(10, 200, (ng, 2))], 1).astype(np.float32)
        pr = g.uniform(0, 300, (npr, 2)); pr = np.concatenate([pr, pr + g.uniform(10, 200, (npr, 2))], 1).astype(np.float32)
        if ng and npr > 3:
            pr[1] = gt[0]                                   # an exact hit
            pr[2] = pr[1]                                   # and a duplicate proposal: IoU tie
            pr[3, :2] = gt[ng - 1, :2] + 3; pr[3, 2:] = gt[ng - 1, 2:] + 3
        sc = g.uniform(0, 1, (npr, 1)).astype(np.float32)
        gts.append(gt if ng else (None if i % 2 else gt))
        props.append(np.concatenate([pr, sc], 1))
    nums, thrs = np.array([1, 5, 20, 100]), np.arange(0.5, 0.96, 0.05)
    fx = {"count": np.asarray(len(gts)), "nums": nums, "thrs": thrs}
    for i, (a, b) in enumerate(zip(gts, props)):
        fx[f"gt{i}"] = np.zeros((0, 4), np.float32) if a is None else a
        fx[f"gt{i}_none"] = np.asarray(a is None)
        fx[f"prop{i}"] = b
    for legacy in (False, True):
        fx[f"recalls_legacy{int(legacy)}"] = rc.eval_recalls(gts, props, nums, thrs, use_legacy_coordinate=legacy)
        # the intermediate the device computes: _recalls' tmp_ious per budget, re-derived through the reference's functions
        sorted_props = [p[np.argsort(p[:, 4])[::-1], :] for p in props]
        all_ious = []
        for a, b in zip(gts, sorted_props):
            pn = min(b.shape[0], nums[-1])
            all_ious.append(np.zeros((0, b.shape[0]), np.float32) if a is None or a.shape[0] == 0
                            else rc.bbox_overlaps(a, b[:pn, :4], use_legacy_coordinate=legacy))
        fx[f"iou0_legacy{int(legacy)}"] = all_ious[0] if all_ious[0].size else np.zeros((0, 0), np.float32)
    np.savez_compressed(os.path.join(OUT, "recall.npz"), **fx)


def case_retrieval_metric():
    """evaluate_retrieval_per_class (retrieval_metric.py:14-47) lifted out of its script (the module body parses
    argv and loads datasets at import) with ast, run on seeded prediction / ground-truth id sets."""
    import ast
    import json
    src = open(os.path.join(REF, "eval_retrieval", "retrieval_metric.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "evaluate_retrieval_per_class")
    ns = {"tqdm": lambda it, **k: it}
    exec(compile(ast.Module(body=[ast.parse("from typing import Dict, List, Set").body[0], fn], type_ignores=[]),
                 "retrieval_metric_fn", "exec"), ns)
    g = np.random.default_rng(31)
    cats = [f"class_{i}" for i in range(12)]
    gt = {c: set(int(v) for v in g.choice(200, int(g.integers(0, 30)), replace=False)) for c in cats}
    pred = {c: [int(v) for v in g.choice(200, int(g.integers(0, 40)), replace=True)] for c in cats[:-2]}
    res = ns["evaluate_retrieval_per_class"](pred, gt)
    blob = json.dumps({"gt": {c: sorted(v) for c, v in gt.items()}, "pred": pred, "result": res}, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "retrieval_metric.npz"), blob=np.asarray(blob))


def _seed_module(mod: torch.nn.Module, seed: int):
    """Deterministic, trained-like parameters for a small reference module (fixture carries them)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in mod.state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            t = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith(("running_mean", "bias")) or k == "bias":
            t = torch.randn(v.shape, generator=g) * 0.2
        elif k.endswith("bn.weight") or k.endswith(".0.weight") and v.dim() == 1:
            t = torch.rand(v.shape, generator=g) + 0.5
        elif k == "scale":
            t = torch.rand(v.shape, generator=g) + 0.25
        else:
            fan = max(1, v[0].numel()) if v.dim() > 1 else 1
            t = torch.randn(v.shape, generator=g) / fan ** 0.5
        out[k] = t.to(torch.float32)
    missing = mod.load_state_dict(out, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    for m in mod.modules():
        m.training = False
    return out


def case_bricks():
    """Text-guided attention bricks (yolo_bricks.py:161-243, 572-648) run as the reference classes; the oracle
    restatement (oracle/bricks.py) must reproduce them bit for bit.  The fixture carries parameters, inputs and
    full outputs (all small)."""
    import json
    from oracle import bricks as obr
    yb = import_yolo_bricks()
    fx = {}
    g = torch.Generator().manual_seed(77)
    msa_cases = [
        dict(in_channels=32, out_channels=64, guide_channels=48, embed_channels=64, num_heads=2, with_scale=False, b=2, h=12, w=9, n=5),
        dict(in_channels=32, out_channels=32, guide_channels=40, embed_channels=32, num_heads=2, with_scale=True, b=1, h=7, w=16, n=80),
        dict(in_channels=64, out_channels=64, guide_channels=24, embed_channels=64, num_heads=1, with_scale=True, b=3, h=5, w=5, n=1),
    ]
    for j, c in enumerate(msa_cases):
        kw = {k: c[k] for k in ("in_channels", "out_channels", "guide_channels", "embed_channels", "num_heads", "with_scale")}
        mod = yb.MaxSigmoidAttnBlock(**kw)
        p = _seed_module(mod, 100 + j)
        x = torch.randn(c["b"], c["in_channels"], c["h"], c["w"], generator=g)
        guide = torch.randn(c["b"], c["n"], c["guide_channels"], generator=g)
        ref = mod(x, guide)
        must_equal(f"bricks.msa{j}", ref, obr.max_sigmoid_attn(x, guide, p, c["num_heads"]))
        fx[f"msa{j}.cfg"] = np.asarray(json.dumps(c))
        fx[f"msa{j}.x"], fx[f"msa{j}.guide"], fx[f"msa{j}.out"] = x.numpy(), guide.numpy(), ref.numpy()
        for k, v in p.items():
            fx[f"msa{j}.p.{k}"] = v.numpy()
    ipa_cases = [
        dict(image_channels=[16, 24, 32], text_channels=48, embed_channels=64, num_heads=4, with_scale=False, b=2, n=5,
             sizes=[[13, 11], [7, 7], [4, 5]]),
        dict(image_channels=[8, 16, 16], text_channels=64, embed_channels=64, num_heads=2, with_scale=True, b=1, n=80,
             sizes=[[9, 9], [6, 3], [3, 3]]),
    ]
    for j, c in enumerate(ipa_cases):
        kw = {k: c[k] for k in ("image_channels", "text_channels", "embed_channels", "num_heads", "with_scale")}
        mod = yb.ImagePoolingAttentionModule(**kw)
        p = _seed_module(mod, 200 + j)
        text = torch.randn(c["b"], c["n"], c["text_channels"], generator=g)
        feats = [torch.randn(c["b"], ch, hh, ww, generator=g) for ch, (hh, ww) in zip(c["image_channels"], c["sizes"])]
        ref = mod(text, feats)
        must_equal(f"bricks.ipa{j}", ref, obr.image_pooling_attention(text, feats, p, c["num_heads"]))
        fx[f"ipa{j}.cfg"] = np.asarray(json.dumps(c))
        fx[f"ipa{j}.text"], fx[f"ipa{j}.out"] = text.numpy(), ref.numpy()
        for l, f in enumerate(feats):
            fx[f"ipa{j}.feat{l}"] = f.numpy()
        for k, v in p.items():
            fx[f"ipa{j}.p.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "bricks.npz"), **fx)


def case_mmdet_geometry():
    """scale_factor / pad_param / shapes produced by the reference's own WeDetectKeepRatioResize + WeDetectLetterResize
    code (transforms.py:62-123, 180-272, 319-330) over image sizes incl. exact fits, up-scaling candidates, odd
    paddings and extreme aspect ratios; mmcv's imresize / impad and mmdet's Resize base are stand-ins (sizes only).
    The product's arithmetic (wedetect_amd.preprocess.mmdet_test_geometry) must reproduce every record."""
    import json
    from wedetect_amd.preprocess import mmdet_test_geometry
    install_stub_finder()
    spec = importlib.util.spec_from_file_location("refwd_transforms", os.path.join(REF, "wedetect", "datasets", "transformers", "transforms.py"))
    tr = importlib.util.module_from_spec(spec)
    sys.modules["refwd_transforms"] = tr
    spec.loader.exec_module(tr)
    g = np.random.default_rng(8)
    sizes = [(720, 1280), (1280, 720), (640, 640), (480, 640), (427, 640), (640, 427), (333, 500), (500, 375), (32, 32),
             (3, 999), (999, 3), (641, 640), (639, 641), (1080, 1920), (3000, 4000), (375, 1242), (100, 37), (1279, 1281)]
    sizes += [(int(g.integers(20, 2500)), int(g.integers(20, 2500))) for _ in range(60)]
    recs = []
    for scale in ((640, 640), (1280, 1280)):
        k = tr.WeDetectKeepRatioResize(scale=scale)
        l = tr.WeDetectLetterResize(scale=scale, allow_scale_up=False, pad_val=dict(img=114))
        for (h, w) in sizes:
            res = dict(img=np.zeros((h, w, 3), np.uint8))
            res = l.transform(k.transform(res))
            mine = mmdet_test_geometry(h, w, scale)
            rec = dict(h=h, w=w, scale=list(scale), img_shape=[int(v) for v in res["img_shape"][:2]],
                       scale_factor=[float(res["scale_factor"][0]), float(res["scale_factor"][1])],
                       pad_param=[float(v) for v in res["pad_param"]])
            assert tuple(rec["img_shape"]) == mine["img_shape"], (rec, mine)
            assert rec["scale_factor"] == list(mine["scale_factor"]), (rec, mine)
            assert rec["pad_param"] == [float(v) for v in mine["pad_param"]] and res["pad_param"].dtype == np.float32, (rec, mine)
            recs.append(rec)
    print(f"  [bit-identical] mmdet test-pipeline geometry, {len(recs)} sizes")
    with open(os.path.join(OUT, "mmdet_geometry.json"), "w") as f:
        json.dump(recs, f)


def case_configs():
    """The ``model`` dict, ``img_scale`` and ``test_pipeline`` of config/wedetect_{tiny,base,large}.py as data (the
    files are plain Python: executed, not copied), for the config-driven builder's tests."""
    import json
    out = {}
    for size in ("tiny", "base", "large"):
        ns = {}
        exec(compile(open(os.path.join(REF, "config", f"wedetect_{size}.py")).read(), f"wedetect_{size}.py", "exec"), ns)
        out[size] = {"model": ns["model"], "img_scale": list(ns["img_scale"]), "test_pipeline": ns["test_pipeline"]}
    with open(os.path.join(OUT, "model_cfgs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference tree not present: goldens can only be generated in the build container"
    if "--only-bricks" in sys.argv:
        case_bricks()
        sys.exit(0)
    if "--only-network" in sys.argv:
        gp = import_generate_proposal()
        case_network(gp, "base", 1, 64)
        case_network(gp, "base", 2, 128)
        case_network(gp, "base", 1, 640)
        case_network(gp, "large", 1, 640, k_text=1203)
        sys.exit(0)
    if "--search-robust" in sys.argv:
        i = sys.argv.index("--search-robust")
        print(search_robust(sys.argv[i + 1], int(sys.argv[i + 2])))
        sys.exit(0)
    if "--only-robust2" in sys.argv:
        gp = import_generate_proposal()
        for arch in ("base", "large"):
            s_mm, s_uni = ROBUST_SEEDS_2[arch]
            k_text = 80 if arch == "base" else 1203
            case_network(gp, arch, 1, 640, seed_img=s_mm, k_text=k_text, out_tag=f"{arch}_b1_640_robust2_mm")
            case_network(gp, arch, 1, 640, seed_img=s_uni, k_text=k_text, out_tag=f"{arch}_b1_640_robust2_uni")
        sys.exit(0)
    if "--only-robust" in sys.argv:
        gp = import_generate_proposal()
        i = sys.argv.index("--only-robust")
        for arch in sys.argv[i + 1:] or ["base", "large"]:
            case_robust(gp, arch, 80 if arch == "base" else 1203)
        sys.exit(0)
    if "--only-tiny-cfg0" in sys.argv:
        # plugin files first (they import transformers), then the torchvision stand-in of generate_proposal
        import_wedetect_models()
        gp = import_generate_proposal()
        case_tiny_config0(gp)
        sys.exit(0)
    if "--only-nms" in sys.argv:
        case_nms()
        sys.exit(0)
    if "--only-configs" in sys.argv:
        case_configs()
        sys.exit(0)
    if "--only-geometry" in sys.argv:
        case_mmdet_geometry()
        sys.exit(0)
    # transformers probes torchvision at import: the plugin files (which import it) must be
    # loaded BEFORE the bare torchvision stand-in goes into sys.modules
    case_mmdet_modules("tiny", 64)
    gp = import_generate_proposal()
    case_tiny_config0(gp)
    case_filter_topk(gp)
    case_nms()
    case_retrieval()
    case_network(gp, "base", 1, 64)
    case_network(gp, "base", 2, 128)
    case_network(gp, "base", 1, 640)
    case_network(gp, "large", 1, 64, full_predict=False)
    case_network(gp, "large", 1, 640, k_text=1203)
    case_robust(gp, "base", 80)
    case_robust(gp, "large", 1203)
    case_letterbox(gp)
    case_recall()
    case_retrieval_metric()
    case_bricks()
    case_configs()
    case_mmdet_geometry()
    print("all golden fixtures written to", OUT)
