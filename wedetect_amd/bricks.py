"""Text-guided attention bricks on the device.

Operator-surface mirrors of the reference classes in wedetect/models/layers/yolo_bricks.py:

  * ``MaxSigmoidAttnBlock`` (161-243)          forward(x [B, C, H, W], guide [B, N, G]) -> [B, out, H, W]
  * ``ImagePoolingAttentionModule`` (572-648)  forward(text [B, N, Ct], [level feature maps]) -> [B, N, Ct]

Same constructor keywords, same state-dict names (``embed_conv.conv.weight``, ``project_conv.bn.*``,
``guide_fc.*``, ``projections.l.conv.*``, ``query.0/1.*`` ...), eval-mode semantics (BatchNorm folded
into its conv at load).  The shipped WeDetect configs set ``mm_neck=False`` (config/wedetect_*.py:40-41),
so these modules are not on the measured path; they exist so that a text-guided neck built on this
package has every brick.  Convolutions / Linear layers run on wd_conv_gemm(_split), LayerNorms on
wd_layernorm_rows, the rest on the three kernels of csrc/bricks.hip.  Feature maps are NHWC rows on the
device; the NCHW tensors of the reference interface are converted at the boundary (``forward_nhwc``
skips that).  There is no CPU path."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import lib as L
from .pack import _conv_rows, _fold_bn


def _np(v) -> np.ndarray:
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


class _Dense:
    """A conv / linear as one GEMM launch: fp32 rows [n][k] (+ lazily split fp16x3 copy) and bias."""

    def __init__(self, w_rows: np.ndarray, bias: np.ndarray, device, precision: str, kh: int = 1, pad: int = 0):
        self.w = torch.from_numpy(np.ascontiguousarray(w_rows, dtype=np.float32)).to(device)
        self.b = torch.from_numpy(np.ascontiguousarray(bias, dtype=np.float32)).to(device)
        self.n, self.k = self.w.shape
        self.kh, self.pad = kh, pad
        self.ws = L.split_weights(self.w) if precision == "fp16x3" and (self.k // (kh * kh)) % 8 == 0 else None

    def __call__(self, a: torch.Tensor, c: torch.Tensor, batch: int, h: int, w: int, res: Optional[torch.Tensor] = None):
        """a: rows [batch * h * w, cin] (pitch = a.stride(0)); c: rows [batch * h * w, >= n]."""
        kw = dict(batch=batch, hin=h, win=w, cin=self.k // (self.kh * self.kh), lda=a.stride(0), kh=self.kh, kw=self.kh,
                  pad=self.pad, n=self.n, ldc=c.stride(0))
        if res is not None:
            kw.update(res=res, ldres=res.stride(0))
        if self.ws is not None:
            L.conv_gemm(a, None, self.b, c, w_split=self.ws, **kw)
        else:
            L.conv_gemm(a, self.w, self.b, c, **kw)


def _conv_module(sd: Dict[str, np.ndarray], name: str, eps: float, device, precision, pad: int = 0) -> _Dense:
    """mmcv ConvModule (conv -> BN, no activation) folded into one GEMM."""
    w = _np(sd[name + ".conv.weight"])
    cb = _np(sd[name + ".conv.bias"]) if name + ".conv.bias" in sd else None
    if name + ".bn.weight" in sd:
        w, b = _fold_bn(w, cb, _np(sd[name + ".bn.weight"]), _np(sd[name + ".bn.bias"]), _np(sd[name + ".bn.running_mean"]),
                        _np(sd[name + ".bn.running_var"]), eps)
    else:
        b = np.zeros(w.shape[0]) if cb is None else cb
    return _Dense(_conv_rows(w), b, device, precision, kh=w.shape[2], pad=pad)


def _rows_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B, C, H, W] (any memory format) -> contiguous NHWC rows [B * H * W, C]."""
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).contiguous().view(b * h * w, c)


class _Brick:
    PRECISIONS = ("fp32", "fp16x3")

    def _init_common(self, precision, device):
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {self.PRECISIONS}")
        self.precision, self.dev = precision, torch.device(device)
        self.training = False
        self._ready = False

    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def _need(self):
        if not self._ready:
            raise RuntimeError("load_state_dict() must be called before forward")

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


class MaxSigmoidAttnBlock(_Brick):
    """yolo_bricks.py:161-243.  ``use_depthwise=True`` (DepthwiseSeparableConvModule projection) is not built."""

    def __init__(self, in_channels: int, out_channels: int, guide_channels: int, embed_channels: int, kernel_size: int = 3,
                 padding: int = 1, num_heads: int = 1, use_depthwise: bool = False, with_scale: bool = False, conv_cfg=None,
                 norm_cfg=dict(type="BN", momentum=0.03, eps=0.001), init_cfg=None, use_einsum: bool = True,
                 precision: str = "fp16x3", device="cuda"):
        assert out_channels % num_heads == 0 and embed_channels % num_heads == 0, \
            "out_channels and embed_channels should be divisible by num_heads."          # :180-182
        if use_depthwise:
            raise NotImplementedError("use_depthwise=True is not supported")
        self._init_common(precision, device)
        self.in_channels, self.out_channels, self.guide_channels = in_channels, out_channels, guide_channels
        self.embed_channels, self.kernel_size, self.padding = embed_channels, kernel_size, padding
        self.num_heads, self.with_scale = num_heads, with_scale
        self.head_channels = out_channels // num_heads                                    # :184
        self.has_embed_conv = embed_channels != in_channels                               # :187-193
        self.bn_eps = float(norm_cfg.get("eps", 1e-5)) if norm_cfg else None

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        eps = self.bn_eps if self.bn_eps is not None else 1e-5
        self.embed_conv = _conv_module(sd, "embed_conv", eps, self.dev, self.precision) if self.has_embed_conv else None
        self.project_conv = _conv_module(sd, "project_conv", eps, self.dev, self.precision, pad=self.padding)
        self.guide_fc = _Dense(_np(sd["guide_fc.weight"]), _np(sd["guide_fc.bias"]), self.dev, self.precision)
        if (self.project_conv.n, self.guide_fc.n) != (self.out_channels, self.embed_channels):
            raise RuntimeError("state dict does not match the constructor's channel counts")
        f = lambda v: torch.from_numpy(np.ascontiguousarray(_np(v), dtype=np.float32).reshape(-1)).to(self.dev)
        self.bias = f(sd["bias"])
        self.scale = f(sd["scale"]) if self.with_scale else None
        self._ready = True
        return self

    @torch.no_grad()
    def forward_nhwc(self, x_rows: torch.Tensor, batch: int, h: int, w: int, guide: torch.Tensor) -> torch.Tensor:
        """x_rows [B * H * W, C] fp32 device rows -> [B * H * W, out] rows."""
        self._need()
        hc = self.head_channels
        if self.embed_channels != self.num_heads * hc:
            # the reference's reshape at :219-221 fails in the same situation
            raise RuntimeError(f"embed_channels {self.embed_channels} cannot be viewed as {self.num_heads} heads of {hc}")
        n_guide = guide.shape[1]
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        g_in = guide.to(self.dev, torch.float32).contiguous().view(batch * n_guide, self.guide_channels)
        g = f(batch * n_guide, self.embed_channels)
        self.guide_fc(g_in, g, 1, 1, batch * n_guide)                                     # :218
        if self.embed_conv is not None:
            e = f(batch * h * w, self.embed_channels)
            self.embed_conv(x_rows, e, batch, h, w)                                       # :220
        else:
            e = x_rows
        y = f(batch * h * w, self.out_channels)
        self.project_conv(x_rows, y, batch, h, w)                                         # :240
        L.max_sigmoid_attn(e, g, self.bias, self.scale, y, batch, h * w, n_guide, self.num_heads, hc, hc)   # :224-242
        return y

    def forward(self, x: torch.Tensor, guide: torch.Tensor) -> torch.Tensor:
        b, _, h, w = x.shape
        y = self.forward_nhwc(_rows_nhwc(L._f32(x, "x")), b, h, w, guide)
        return y.view(b, h, w, self.out_channels).permute(0, 3, 1, 2)                     # NCHW view of the NHWC result


class ImagePoolingAttentionModule(_Brick):
    """yolo_bricks.py:572-648: the text rows attend over 3 x 3 max-pooled patches of every level."""

    def __init__(self, image_channels: List[int], text_channels: int, embed_channels: int, with_scale: bool = False,
                 num_feats: int = 3, num_heads: int = 8, pool_size: int = 3, use_einsum: bool = True,
                 precision: str = "fp16x3", device="cuda"):
        self._init_common(precision, device)
        self.image_channels, self.text_channels, self.embed_channels = list(image_channels), text_channels, embed_channels
        self.with_scale, self.num_feats, self.num_heads, self.pool_size = with_scale, num_feats, num_heads, pool_size
        self.head_channels = embed_channels // num_heads                                  # :592

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items()}
        self.projections = [_conv_module(sd, f"projections.{l}", 1e-5, self.dev, self.precision)
                            for l in range(len(self.image_channels))]
        f = lambda v: torch.from_numpy(np.ascontiguousarray(_np(v), dtype=np.float32).reshape(-1)).to(self.dev)
        self.ln, self.fc = {}, {}
        for name in ("query", "key", "value"):
            self.ln[name] = (f(sd[name + ".0.weight"]), f(sd[name + ".0.bias"]))
            self.fc[name] = _Dense(_np(sd[name + ".1.weight"]), _np(sd[name + ".1.bias"]), self.dev, self.precision)
        pw, pb = _np(sd["proj.weight"]).astype(np.float64), _np(sd["proj.bias"]).astype(np.float64)
        if self.with_scale:
            s = float(_np(sd["scale"]).reshape(-1)[0])       # x * scale + text (:648): the scalar is folded into proj
            pw, pb = pw * s, pb * s
        self.proj = _Dense(pw, pb, self.dev, self.precision)
        self._ready = True
        return self

    @torch.no_grad()
    def forward(self, text_features: torch.Tensor, image_features: Sequence[torch.Tensor]) -> torch.Tensor:
        self._need()
        assert len(image_features) == self.num_feats                                      # :616
        b = image_features[0].shape[0]
        p2, e_ch, hc = self.pool_size ** 2, self.embed_channels, self.head_channels
        n_k = self.num_feats * p2
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        patches = f(b * n_k, e_ch)
        for l, (x, proj) in enumerate(zip(image_features, self.projections)):            # :618-624
            _, _, h, w = x.shape
            y = f(b * h * w, e_ch)
            proj(_rows_nhwc(L._f32(x, "image_features")), y, b, h, w)
            L.adaptive_maxpool_nhwc(y, patches[l * p2:], n_k * e_ch, b, h, w, e_ch, self.pool_size)
        text = L._f32(text_features, "text_features").contiguous()
        n_q = text.shape[1]
        t_rows = text.view(b * n_q, self.text_channels)

        def ln_fc(rows, name, d):
            normed = f(rows.shape[0], d)
            L.layernorm_rows(rows, normed, *self.ln[name], rows.shape[0], d, eps=1e-5)
            out = f(rows.shape[0], e_ch)
            self.fc[name](normed, out, 1, 1, rows.shape[0])
            return out
        q = ln_fc(t_rows, "query", self.text_channels)                                    # :625
        k = ln_fc(patches, "key", e_ch)                                                   # :626
        v = ln_fc(patches, "value", e_ch)                                                 # :627
        att = f(b * n_q, e_ch)
        L.cross_attention_small(q, k, v, att, b, n_q, n_k, self.num_heads, hc)            # :629-646
        out = f(b * n_q, self.text_channels)
        self.proj(att, out, 1, 1, b * n_q, res=t_rows)                                    # :647-648
        return out.view(b, n_q, self.text_channels)
