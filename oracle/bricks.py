"""TEST INFRASTRUCTURE — plain-torch restatement (CPU, fp32) of the reference's two text-guided
attention bricks:

  * ``MaxSigmoidAttnBlock.forward``          wedetect/models/layers/yolo_bricks.py:214-243
  * ``ImagePoolingAttentionModule.forward``  wedetect/models/layers/yolo_bricks.py:614-648

Both take a flat ``{name: tensor}`` dict with the reference modules' own state-dict names.  The
reference builds its convolutions with ``mmcv.cnn.ConvModule`` (third party, absent here); its
documented behaviour — ``Conv2d(bias = no norm) -> BatchNorm2d(eps from norm_cfg) -> activation`` —
is what ``conv_module`` restates.  Pinned: tests/golden/make_golden.py runs the reference classes
themselves (over a ConvModule / Linear stand-in with exactly those semantics) on seeded inputs and
aborts unless these functions reproduce them bit for bit; tests/test_cpu.py re-checks against the
committed fixture.  Only tests/, smoke() and bench.py's cpu_baseline may import this module."""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def conv_module(x: torch.Tensor, p: Dict[str, torch.Tensor], name: str, padding: int = 0, bn_eps: float = 1e-3) -> torch.Tensor:
    """mmcv ConvModule with ``act_cfg=None``: conv (+ bias when there is no norm layer) then eval-mode BN."""
    y = F.conv2d(x, p[name + ".conv.weight"], p.get(name + ".conv.bias"), 1, padding)
    if name + ".bn.weight" in p:
        y = F.batch_norm(y, p[name + ".bn.running_mean"], p[name + ".bn.running_var"], p[name + ".bn.weight"],
                         p[name + ".bn.bias"], False, 0.0, bn_eps)
    return y


def max_sigmoid_attn(x: torch.Tensor, guide: torch.Tensor, p: Dict[str, torch.Tensor], num_heads: int, padding: int = 1,
                     bn_eps: float = 1e-3) -> torch.Tensor:
    """x [B, C, H, W], guide [B, N, G] -> [B, out, H, W]   (yolo_bricks.py:214-243, einsum branch)."""
    b, _, h, w = x.shape
    out_channels = p["project_conv.conv.weight"].shape[0]
    hc = out_channels // num_heads                                           # :199
    g = F.linear(guide, p["guide_fc.weight"], p["guide_fc.bias"])            # :218
    g = g.reshape(b, -1, num_heads, hc)                                      # :219
    e = conv_module(x, p, "embed_conv", 0, bn_eps) if "embed_conv.conv.weight" in p else x     # :220
    e = e.reshape(b, num_heads, hc, h, w)                                    # :221
    a = torch.einsum("bmchw,bnmc->bmhwn", e, g)                              # :224
    a = a.max(dim=-1)[0]                                                     # :235
    a = a / (hc ** 0.5)                                                      # :236
    a = a + p["bias"][None, :, None, None]                                   # :237
    scale = p["scale"] if "scale" in p else 1.0                              # :204-207
    a = a.sigmoid() * scale                                                  # :238
    y = conv_module(x, p, "project_conv", padding, bn_eps)                   # :240
    y = y.reshape(b, num_heads, -1, h, w) * a.unsqueeze(2)                   # :241-242
    return y.reshape(b, -1, h, w)                                            # :243


def image_pooling_attention(text: torch.Tensor, feats: Sequence[torch.Tensor], p: Dict[str, torch.Tensor], num_heads: int,
                            pool_size: int = 3) -> torch.Tensor:
    """text [B, N, Ct], feats = per-level [B, C_l, H_l, W_l] -> [B, N, Ct]   (yolo_bricks.py:614-648)."""
    b = feats[0].shape[0]
    e_ch = p["proj.weight"].shape[1]
    hc = e_ch // num_heads                                                   # :592
    patches = []
    for l, x in enumerate(feats):                                            # :618-622
        y = conv_module(x, p, f"projections.{l}")
        patches.append(F.adaptive_max_pool2d(y, (pool_size, pool_size)).view(b, -1, pool_size * pool_size))
    m = torch.cat(patches, dim=-1).transpose(1, 2)                           # :623-624  [B, L*P*P, E]

    def ln_fc(t, name):
        d = t.shape[-1]
        return F.linear(F.layer_norm(t, (d,), p[name + ".0.weight"], p[name + ".0.bias"], 1e-5),
                        p[name + ".1.weight"], p[name + ".1.bias"])
    q = ln_fc(text, "query").reshape(b, -1, num_heads, hc)                   # :625, 629
    k = ln_fc(m, "key").reshape(b, -1, num_heads, hc)                        # :626, 630
    v = ln_fc(m, "value").reshape(b, -1, num_heads, hc)                      # :627, 631
    a = torch.einsum("bnmc,bkmc->bmnk", q, k)                                # :633
    a = F.softmax(a / (hc ** 0.5), dim=-1)                                   # :639-640
    o = torch.einsum("bmnk,bkmc->bnmc", a, v)                                # :642
    o = F.linear(o.reshape(b, -1, e_ch), p["proj.weight"], p["proj.bias"])   # :647
    scale = p["scale"] if "scale" in p else 1.0                              # :594-597
    return o * scale + text                                                  # :648
