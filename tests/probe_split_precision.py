"""Error-budget probe (CPU, not collected by pytest): what do split low-precision MFMA inputs cost?

Runs the oracle network with every dense conv / linear evaluated as
  * fp32 (the product path today),
  * "bf16x3":  x = xh + xl (bf16 each), w likewise;  x.w ~ xh.wh + xh.wl + xl.wh, fp32 accumulate,
  * "fp16x3":  the same with fp16 halves (inputs pre-scaled by a power of two per tensor so the
               low halves stay out of the fp16 subnormal range),
  * "bf16":    plain bf16 inputs,
and compares embeddings / scores with a float64 run of the same network.  Depthwise convs, LN,
activations stay fp32 in all variants (they are not MFMA work).

    python -m tests.probe_split_precision [arch] [size] [batch]
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_cpu as orc
from wedetect_amd import weights as W
from wedetect_amd.arch import get_arch


def _split(x, dt):
    hi = x.to(dt).to(torch.float32)
    lo = (x - hi).to(dt).to(torch.float32)
    return hi, lo


def _pow2_scale(x, target=1024.0):
    m = float(x.abs().max())
    if m == 0.0:
        return 1.0
    return float(2.0 ** np.floor(np.log2(target / m)))


class SplitF:
    """Stand-in for torch.nn.functional inside oracle.ref_cpu with split-precision dense ops."""

    def __init__(self, mode):
        self.mode = mode

    def __getattr__(self, name):
        return getattr(F, name)

    def _dense(self, op, x, w, bias, **kw):
        if self.mode == "fp32" or kw.get("groups", 1) != 1:
            return op(x, w, bias, **kw)
        if self.mode == "bf16":
            y = op(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), None, **kw)
        else:
            dt = torch.bfloat16 if self.mode == "bf16x3" else torch.float16
            sx = sw = 1.0
            if dt is torch.float16:
                sx, sw = _pow2_scale(x), _pow2_scale(w)
            xh, xl = _split(x * sx, dt)
            wh, wl = _split(w * sw, dt)
            y = op(xh, wh, None, **kw) + (op(xh, wl, None, **kw) + op(xl, wh, None, **kw))
            y = y / (sx * sw)
        if bias is not None:
            shape = [1, -1] + [1] * (y.dim() - 2) if op is not F.linear else [-1]
            y = y + bias.view(shape)
        return y

    def conv2d(self, x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        return self._dense(F.conv2d, x, w, bias, stride=stride, padding=padding, groups=groups)

    def linear(self, x, w, bias=None):
        return self._dense(F.linear, x, w, bias)

    def conv_transpose2d(self, x, w, bias=None, stride=1, padding=0):
        return self._dense(F.conv_transpose2d, x, w, bias, stride=stride, padding=padding)


def run(sd, arch, imgs, bank, mode, dtype=torch.float32):
    old = orc.F
    orc.F = SplitF(mode)
    try:
        sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        with torch.no_grad():
            x = orc.preprocess_u8(imgs).to(dtype)
            c = orc.backbone(sdd, arch, x)
            p = orc.neck(sdd, arch, c)
            embs, logits = [], []
            for l in range(3):
                e, lg, _ = orc.head_level(sdd, l, p[l], bank.to(dtype), True) if dtype == torch.float32 else \
                    _head64(sdd, l, p[l], bank.to(dtype))
                embs.append(e.flatten(2))
                logits.append(lg.flatten(2))
        return torch.cat(embs, 2).double(), torch.cat(logits, 2).double()
    finally:
        orc.F = old


def _head64(sd, l, feat, text):
    embed = orc._head_branch(sd, orc.HD + f"cls_preds.{l}", feat)
    q = orc.HD + f"cls_contrasts.{l}"
    embed = F.batch_norm(embed, sd[q + ".norm.running_mean"], sd[q + ".norm.running_var"],
                         sd[q + ".norm.weight"], sd[q + ".norm.bias"], False, 0.03, 1e-3)
    t = F.normalize(text, dim=-1, p=2)
    logits = torch.einsum("bchw,kc->bkhw", embed, t) * sd[q + ".logit_scale"].exp() + sd[q + ".bias"]
    return embed, logits, None


def main():
    arch_name = sys.argv[1] if len(sys.argv) > 1 else "base"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    arch = get_arch(arch_name)
    sd = orc.to_torch(W.make_state_dict(arch_name))
    imgs = W.make_images(batch, size, size)
    bank = torch.from_numpy(W.make_text_bank(80))
    e64, l64 = run(sd, arch, imgs, bank, "fp32", torch.float64)
    s64 = torch.sigmoid(l64)
    print(f"{arch_name} {size}x{size} B={batch}: |embedding| max {float(e64.abs().max()):.2f} rms {float(e64.pow(2).mean().sqrt()):.3f}")
    print(f"{'mode':8s} {'max|d emb|':>12s} {'rms d emb':>12s} {'max|d logit|':>13s} {'max|d score|':>13s}")
    for mode in ("fp32", "fp16x3", "bf16x3", "bf16"):
        e, lg = run(sd, arch, imgs, bank, mode)
        print(f"{mode:8s} {float((e - e64).abs().max()):12.3e} {float((e - e64).pow(2).mean().sqrt()):12.3e} "
              f"{float((lg - l64).abs().max()):13.3e} {float((torch.sigmoid(lg) - s64).abs().max()):13.3e}")


if __name__ == "__main__":
    main()
