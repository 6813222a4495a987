"""Weight pre-pack: mmdet-named state dict -> device tensors in the layouts the kernels read.

Done once at load time, in float64 then rounded to fp32:
  * every eval-mode BatchNorm is folded into the conv that feeds it
      neck ConvModule_torch: eps 1e-5 (nn.BatchNorm2d default, yolo_world_pafpn.py:55)
      head ConvModule + BNContrastiveHead.norm: eps 1e-3 (yolov8_head.py:54-56, yolo_world_head.py:83)
  * ConvNeXt layer-scale gamma is folded into pwconv2 (mm_backbone.py:119-121)
  * conv weights OIHW -> [O][(kh, kw, I)] rows (k contiguous, matches the NHWC im2col order)
  * depthwise 7x7 [C,1,7,7] -> [49][C]
  * ConvTranspose2d [I,O,2,2] -> [(ty, tx, O)][I], bias repeated per tap
nn.Linear weights are already [out][in] = [n][k] and are used as they are.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .arch import ArchSpec, BB, HD, NK, get_arch


def _dev(a: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def _conv_rows(w: np.ndarray) -> np.ndarray:
    """OIHW -> [O, kh*kw*I] with (kh, kw, i) order."""
    o = w.shape[0]
    return np.transpose(w, (0, 2, 3, 1)).reshape(o, -1)


def _fold_bn(w: np.ndarray, conv_bias, bn_w, bn_b, bn_m, bn_v, eps: float):
    s = bn_w.astype(np.float64) / np.sqrt(bn_v.astype(np.float64) + eps)
    wf = w.astype(np.float64) * s.reshape(-1, *([1] * (w.ndim - 1)))
    cb = np.zeros_like(s) if conv_bias is None else conv_bias.astype(np.float64)
    bf = (cb - bn_m.astype(np.float64)) * s + bn_b.astype(np.float64)
    return wf, bf


class Packed:
    """Flat namespace of device tensors + python scalars, keyed by short names."""

    def __init__(self):
        self.t: Dict[str, torch.Tensor] = {}
        self.s: Dict[str, float] = {}

    def __getitem__(self, k):
        return self.t[k]


def pack(sd: Dict[str, np.ndarray], arch, device="cuda") -> Packed:
    a: ArchSpec = get_arch(arch) if isinstance(arch, str) else arch
    P = Packed()
    g = lambda k: np.asarray(sd[k])

    # ---------------------------------------------------------------- backbone
    d = BB + "downsample_layers."
    P.t["stem.w"] = _dev(_conv_rows(g(d + "0.0.weight")), device)          # [C0, 48]
    P.t["stem.b"] = _dev(g(d + "0.0.bias"), device)
    P.t["stem.ln_w"] = _dev(g(d + "0.1.weight"), device)
    P.t["stem.ln_b"] = _dev(g(d + "0.1.bias"), device)
    for i in range(1, 4):
        P.t[f"down{i}.ln_w"] = _dev(g(d + f"{i}.0.weight"), device)
        P.t[f"down{i}.ln_b"] = _dev(g(d + f"{i}.0.bias"), device)
        P.t[f"down{i}.w"] = _dev(_conv_rows(g(d + f"{i}.1.weight")), device)   # [Co, 4*Ci]
        P.t[f"down{i}.b"] = _dev(g(d + f"{i}.1.bias"), device)
    for i in range(4):
        c = a.dims[i]
        for j in range(a.depths[i]):
            p = BB + f"stages.{i}.{j}."
            q = f"s{i}.{j}."
            P.t[q + "dw_w"] = _dev(g(p + "dwconv.weight").reshape(c, 49).T, device)   # [49, C]
            P.t[q + "dw_b"] = _dev(g(p + "dwconv.bias"), device)
            P.t[q + "ln_w"] = _dev(g(p + "norm.weight"), device)
            P.t[q + "ln_b"] = _dev(g(p + "norm.bias"), device)
            P.t[q + "w1"] = _dev(g(p + "pwconv1.weight"), device)                    # [4C, C]
            P.t[q + "b1"] = _dev(g(p + "pwconv1.bias"), device)
            gm = g(p + "gamma").astype(np.float64)
            P.t[q + "w2"] = _dev(g(p + "pwconv2.weight").astype(np.float64) * gm[:, None], device)   # [C, 4C]
            P.t[q + "b2"] = _dev(g(p + "pwconv2.bias").astype(np.float64) * gm, device)

    # ---------------------------------------------------------------- neck
    def neck_conv(name: str, short: str):
        p = NK + name + ".block."
        wf, bf = _fold_bn(g(p + "conv.weight"), None, g(p + "bn.weight"), g(p + "bn.bias"),
                          g(p + "bn.running_mean"), g(p + "bn.running_var"), 1e-5)
        P.t[short + ".w"] = _dev(_conv_rows(wf), device)
        P.t[short + ".b"] = _dev(bf, device)

    def bottlerep(name: str, short: str):
        neck_conv(name + ".conv1", short + ".c1")
        neck_conv(name + ".conv2", short + ".c2")
        P.s[short + ".alpha"] = float(g(NK + name + ".alpha").reshape(-1)[0])

    def bepc3(name: str):
        for cv in ("cv1", "cv2", "cv3"):
            neck_conv(f"{name}.{cv}", f"{name}.{cv}")
        bottlerep(f"{name}.m.conv1", f"{name}.m0")
        for j in range(a.neck_repeats // 2 - 1):
            bottlerep(f"{name}.m.block.{j}", f"{name}.m{j + 1}")

    def bifusion(name: str):
        for cv in ("cv1", "cv2", "cv3", "downsample"):
            neck_conv(f"{name}.{cv}", f"{name}.{cv}")
        w = g(NK + name + ".upsample.upsample_transpose.weight")        # [I, O, 2, 2]
        ci, co = w.shape[0], w.shape[1]
        P.t[name + ".up.w"] = _dev(np.transpose(w, (2, 3, 1, 0)).reshape(4 * co, ci), device)
        P.t[name + ".up.b"] = _dev(np.tile(g(NK + name + ".upsample.upsample_transpose.bias"), 4), device)

    for nm in ("reduce_layer0", "reduce_layer1", "downsample1", "downsample2"):
        neck_conv(nm, nm)
    bifusion("Bifusion0")
    bifusion("Bifusion1")
    for nm in ("Rep_p4", "Rep_p3", "Rep_n3", "Rep_n4"):
        bepc3(nm)

    # ---------------------------------------------------------------- head
    for l in range(3):
        for br in ("cls", "reg"):
            p = HD + f"{br}_preds.{l}"
            for s in ("0", "1"):
                wf, bf = _fold_bn(g(f"{p}.{s}.conv.weight"), None, g(f"{p}.{s}.bn.weight"), g(f"{p}.{s}.bn.bias"),
                                  g(f"{p}.{s}.bn.running_mean"), g(f"{p}.{s}.bn.running_var"), 1e-3)
                P.t[f"head{l}.{br}{s}.w"] = _dev(_conv_rows(wf), device)
                P.t[f"head{l}.{br}{s}.b"] = _dev(bf, device)
        q = HD + f"cls_contrasts.{l}"
        # final 1x1 conv of the cls branch with the contrastive head's BatchNorm folded in:
        # the kernel's output IS the post-BN region embedding (generate_proposal.py:1128-1129)
        wf, bf = _fold_bn(g(HD + f"cls_preds.{l}.2.weight"), g(HD + f"cls_preds.{l}.2.bias"), g(q + ".norm.weight"),
                          g(q + ".norm.bias"), g(q + ".norm.running_mean"), g(q + ".norm.running_var"), 1e-3)
        P.t[f"head{l}.embed.w"] = _dev(_conv_rows(wf), device)
        P.t[f"head{l}.embed.b"] = _dev(bf, device)
        P.t[f"head{l}.dist.w"] = _dev(_conv_rows(g(HD + f"reg_preds.{l}.2.weight")), device)
        P.t[f"head{l}.dist.b"] = _dev(g(HD + f"reg_preds.{l}.2.bias"), device)
        P.s[f"head{l}.logit_scale"] = float(np.asarray(g(q + ".logit_scale")).reshape(-1)[0])
        P.s[f"head{l}.bias"] = float(np.asarray(g(q + ".bias")).reshape(-1)[0])
    if "embeddings" in sd:
        P.t["prompts"] = _dev(g("embeddings"), device)
    return P
