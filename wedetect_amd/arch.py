"""Architecture tables and the state-dict layout (the weight ABI) of the WeDetect
image tower, neck and head.

Follows the reference's constructors:
  * ConvNeXt depths/dims        — wedetect/models/backbones/mm_backbone.py:281-288
  * CSPRepBiFPANNeck channels   — wedetect/models/necks/yolo_world_pafpn.py:999-1082
  * YOLOWorldHeadModule widths  — wedetect/models/dense_heads/yolo_world_head.py:174-232
  * state-dict key names        — the module attribute names of those classes
    (SURVEY.md §8b "State-dict layout").

Nothing here computes; it only enumerates tensors (name -> shape, role) so that the
weight generator, the packer and the oracle agree on one layout.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

EMBED_DIM = 768          # text/region embedding width (wedetect_base.py:9 text_channels)
REG_MAX = 16             # DFL bins (yolov8_head.py reg_max default)
CLS_MID = 256            # cls branch width  = max(in_channels[0]=256, num_train_classes=80)
REG_MID = 64             # reg branch width  = max(16, 256 // 4, 4 * reg_max)
STRIDES = (8, 16, 32)    # wedetect_base.py:78-79


@dataclass(frozen=True)
class ArchSpec:
    name: str
    depths: Tuple[int, int, int, int]
    dims: Tuple[int, int, int, int]
    neck_scale: float
    neck_repeats: int        # BepC3 ``n`` for Rep_p4/p3/n3/n4
    text_dim: int            # hidden size of the XLM-R tower feeding the 768-d projection

    # ---- derived sizes -------------------------------------------------
    @property
    def neck_channels(self) -> Dict[str, int]:
        cl = [64, 128, 256, 512, 1024, 256, 128, 128, 256, 256, 512]
        s = self.neck_scale
        return {
            "c4": int(cl[4] * s), "c3": int(cl[3] * s), "c2": int(cl[2] * s), "c1": int(cl[1] * s),
            "p5r": int(cl[5] * s),   # reduce_layer0 out / Bifusion0 / Rep_p4
            "p4r": int(cl[6] * s),   # reduce_layer1 out / Bifusion1 / Rep_p3
            "d2": int(cl[7] * s),    # downsample2 out
            "n3": int(cl[8] * s),    # Rep_n3 out
            "d1": int(cl[9] * s),    # downsample1 out
            "n4": int(cl[10] * s),   # Rep_n4 out
        }

    @property
    def head_in(self) -> Tuple[int, int, int]:
        nc = self.neck_channels
        return (nc["p4r"], nc["n3"], nc["n4"])


ARCHS: Dict[str, ArchSpec] = {
    "tiny": ArchSpec("tiny", (3, 3, 9, 3), (96, 192, 384, 768), 0.75, 6, 768),
    "base": ArchSpec("base", (3, 3, 27, 3), (128, 256, 512, 1024), 1.0, 12, 768),
    "large": ArchSpec("large", (3, 3, 27, 3), (192, 384, 768, 1536), 1.5, 12, 1024),
    # ``nano`` is not a reference model: a 2-block-per-stage miniature with the same
    # topology, used only to keep CPU-side tests and golden fixtures small.
    "nano": ArchSpec("nano", (1, 1, 2, 1), (32, 64, 128, 256), 0.25, 4, 768),
}


def get_arch(name: str) -> ArchSpec:
    try:
        return ARCHS[name]
    except KeyError:
        raise KeyError(f"unknown WeDetect size {name!r}; expected one of {sorted(ARCHS)}")


# --------------------------------------------------------------------------
# state-dict enumeration
# --------------------------------------------------------------------------
# role tags drive the synthetic-weight statistics (weights.py) — they are not
# part of the reference.
ParamList = List[Tuple[str, Tuple[int, ...], str]]

BB = "backbone.image_model.model."
NK = "neck."
HD = "bbox_head.head_module."


def _conv_bn(prefix: str, cin: int, cout: int, k: int) -> ParamList:
    """``ConvBNReLU/ConvBNSiLU(...).block`` = ConvModule_torch(conv(no bias) + bn)."""
    return [
        (prefix + ".block.conv.weight", (cout, cin, k, k), "conv"),
        (prefix + ".block.bn.weight", (cout,), "bn_w"),
        (prefix + ".block.bn.bias", (cout,), "bn_b"),
        (prefix + ".block.bn.running_mean", (cout,), "bn_m"),
        (prefix + ".block.bn.running_var", (cout,), "bn_v"),
    ]


def _head_conv_bn(prefix: str, cin: int, cout: int) -> ParamList:
    """mmcv ConvModule(conv 3x3 no bias, bn, SiLU) -> keys ``.conv.weight``, ``.bn.*``."""
    return [
        (prefix + ".conv.weight", (cout, cin, 3, 3), "conv"),
        (prefix + ".bn.weight", (cout,), "bn_w"),
        (prefix + ".bn.bias", (cout,), "bn_b"),
        (prefix + ".bn.running_mean", (cout,), "bn_m"),
        (prefix + ".bn.running_var", (cout,), "bn_v"),
    ]


def _bottlerep(prefix: str, c: int) -> ParamList:
    out: ParamList = []
    out += _conv_bn(prefix + ".conv1", c, c, 3)
    out += _conv_bn(prefix + ".conv2", c, c, 3)
    out.append((prefix + ".alpha", (1,), "alpha"))
    return out


def _bepc3(prefix: str, cin: int, cout: int, n: int) -> ParamList:
    c_ = int(cout * 0.5)
    out: ParamList = []
    out += _conv_bn(prefix + ".cv1", cin, c_, 1)
    out += _conv_bn(prefix + ".cv2", cin, c_, 1)
    out += _conv_bn(prefix + ".cv3", 2 * c_, cout, 1)
    out += _bottlerep(prefix + ".m.conv1", c_)
    for j in range(n // 2 - 1):
        out += _bottlerep(prefix + f".m.block.{j}", c_)
    return out


def _bifusion(prefix: str, cin0: int, cin1: int, cout: int) -> ParamList:
    out: ParamList = []
    out += _conv_bn(prefix + ".cv1", cin0, cout, 1)
    out += _conv_bn(prefix + ".cv2", cin1, cout, 1)
    out += _conv_bn(prefix + ".cv3", cout * 3, cout, 1)
    out.append((prefix + ".upsample.upsample_transpose.weight", (cout, cout, 2, 2), "deconv"))
    out.append((prefix + ".upsample.upsample_transpose.bias", (cout,), "bias"))
    out += _conv_bn(prefix + ".downsample", cout, cout, 3)
    return out


def backbone_params(a: ArchSpec) -> ParamList:
    d = a.dims
    out: ParamList = [
        (BB + "downsample_layers.0.0.weight", (d[0], 3, 4, 4), "conv"),
        (BB + "downsample_layers.0.0.bias", (d[0],), "bias"),
        (BB + "downsample_layers.0.1.weight", (d[0],), "ln_w"),
        (BB + "downsample_layers.0.1.bias", (d[0],), "ln_b"),
    ]
    for i in range(1, 4):
        out += [
            (BB + f"downsample_layers.{i}.0.weight", (d[i - 1],), "ln_w"),
            (BB + f"downsample_layers.{i}.0.bias", (d[i - 1],), "ln_b"),
            (BB + f"downsample_layers.{i}.1.weight", (d[i], d[i - 1], 2, 2), "conv"),
            (BB + f"downsample_layers.{i}.1.bias", (d[i],), "bias"),
        ]
    for i in range(4):
        c = d[i]
        for j in range(a.depths[i]):
            p = BB + f"stages.{i}.{j}."
            out += [
                (p + "dwconv.weight", (c, 1, 7, 7), "dwconv"),
                (p + "dwconv.bias", (c,), "bias"),
                (p + "norm.weight", (c,), "ln_w"),
                (p + "norm.bias", (c,), "ln_b"),
                (p + "pwconv1.weight", (4 * c, c), "linear"),
                (p + "pwconv1.bias", (4 * c,), "bias"),
                (p + "pwconv2.weight", (c, 4 * c), "linear"),
                (p + "pwconv2.bias", (c,), "bias"),
                (p + "gamma", (c,), "layer_scale"),
            ]
    return out


def neck_params(a: ArchSpec) -> ParamList:
    nc = a.neck_channels
    n = a.neck_repeats
    out: ParamList = []
    out += _conv_bn(NK + "reduce_layer0", nc["c4"], nc["p5r"], 1)
    out += _bifusion(NK + "Bifusion0", nc["c3"], nc["c2"], nc["p5r"])
    out += _bepc3(NK + "Rep_p4", nc["p5r"], nc["p5r"], n)
    out += _conv_bn(NK + "reduce_layer1", nc["p5r"], nc["p4r"], 1)
    out += _bifusion(NK + "Bifusion1", nc["c2"], nc["c1"], nc["p4r"])
    out += _bepc3(NK + "Rep_p3", nc["p4r"], nc["p4r"], n)
    out += _conv_bn(NK + "downsample2", nc["p4r"], nc["d2"], 3)
    out += _bepc3(NK + "Rep_n3", nc["p4r"] + nc["d2"], nc["n3"], n)
    out += _conv_bn(NK + "downsample1", nc["n3"], nc["d1"], 3)
    out += _bepc3(NK + "Rep_n4", nc["p5r"] + nc["d1"], nc["n4"], n)
    return out


def head_params(a: ArchSpec) -> ParamList:
    out: ParamList = []
    for l, cin in enumerate(a.head_in):
        p = HD + f"cls_preds.{l}"
        out += _head_conv_bn(p + ".0", cin, CLS_MID)
        out += _head_conv_bn(p + ".1", CLS_MID, CLS_MID)
        out += [(p + ".2.weight", (EMBED_DIM, CLS_MID, 1, 1), "conv"),
                (p + ".2.bias", (EMBED_DIM,), "bias")]
        p = HD + f"reg_preds.{l}"
        out += _head_conv_bn(p + ".0", cin, REG_MID)
        out += _head_conv_bn(p + ".1", REG_MID, REG_MID)
        out += [(p + ".2.weight", (4 * REG_MAX, REG_MID, 1, 1), "conv"),
                (p + ".2.bias", (4 * REG_MAX,), "dfl_bias")]
        p = HD + f"cls_contrasts.{l}"
        out += [(p + ".norm.weight", (EMBED_DIM,), "bn_w"),
                (p + ".norm.bias", (EMBED_DIM,), "bn_b"),
                (p + ".norm.running_mean", (EMBED_DIM,), "bn_m"),
                (p + ".norm.running_var", (EMBED_DIM,), "bn_v"),
                (p + ".bias", (), f"contrast_bias{l}"),
                (p + ".logit_scale", (), f"logit_scale{l}")]
    return out


def all_params(a: ArchSpec, num_prompts: int = 0) -> ParamList:
    """Every tensor of the image-side detector in mmdet checkpoint naming.  With
    ``num_prompts`` > 0 the WeDetect-Uni ``embeddings`` [num_prompts, 768] row bank
    (generate_proposal.py:1076-1078) is appended."""
    out = backbone_params(a) + neck_params(a) + head_params(a)
    if num_prompts:
        out.append(("embeddings", (num_prompts, EMBED_DIM), "prompts"))
    return out


def level_sizes(h: int, w: int) -> List[Tuple[int, int]]:
    """Feature-map (H, W) per head level for an input of h x w (multiples of 32)."""
    if h % 32 or w % 32:
        raise ValueError(f"input size must be a multiple of 32, got {h}x{w}")
    return [(h // s, w // s) for s in STRIDES]


def num_anchors(h: int, w: int) -> int:
    return sum(a * b for a, b in level_sizes(h, w))
