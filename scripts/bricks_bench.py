#!/usr/bin/env python3
"""On-device timing of the row-f4 bricks at YOLO-World-like sizes (not on the WeDetect hot path: mm_neck=False).
Prints per-op time and algorithmic HBM GB/s for the stand-alone kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wedetect_amd import lib as L                                   # noqa: E402
from wedetect_amd.bricks import ImagePoolingAttentionModule, MaxSigmoidAttnBlock   # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3       # us


def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    for (b, hw, c, heads, n) in [(32, 80, 128, 4, 80), (32, 40, 256, 8, 80), (32, 20, 512, 16, 80), (16, 80, 128, 4, 1203)]:
        rows = b * hw * hw
        e = torch.randn(rows, c, generator=g).to(dev)
        x = torch.randn(rows, c, generator=g).to(dev)
        gd = torch.randn(b, n, c, generator=g).to(dev)
        hb = torch.zeros(heads, device=dev)
        us = timed(lambda: L.max_sigmoid_attn(e, gd, hb, None, x, b, hw * hw, n, heads, c // heads, c // heads))
        byt = 3 * rows * c * 4 + gd.numel() * 4
        print(f"max_sigmoid_attn  B={b} {hw}x{hw} C={c} heads={heads} N={n}: {us:8.1f} us  {byt / us / 1e3:7.1f} GB/s algorithmic, "
              f"{2.0 * rows * n * c / us / 1e6:6.2f} TFLOP/s fp32 VALU")
        for prec in ("fp32", "fp16x3"):
            sd = {"guide_fc.weight": torch.randn(c, 512, generator=g) * 0.04, "guide_fc.bias": torch.zeros(c), "bias": torch.zeros(heads),
                  "project_conv.conv.weight": torch.randn(c, c, 3, 3, generator=g) * 0.03}
            for k in ("weight", "running_var"):
                sd[f"project_conv.bn.{k}"] = torch.ones(c)
            for k in ("bias", "running_mean"):
                sd[f"project_conv.bn.{k}"] = torch.zeros(c)
            m = MaxSigmoidAttnBlock(c, c, 512, c, num_heads=heads, precision=prec).load_state_dict(sd)
            g512 = torch.randn(b, n, 512, generator=g).to(dev)
            us = timed(lambda: m.forward_nhwc(x, b, hw, hw, g512), reps=10)
            print(f"  MaxSigmoidAttnBlock.forward_nhwc [{prec}] (guide_fc + 3x3 project_conv + attention): {us:8.1f} us")
    chans, sizes = [128, 256, 512], [80, 40, 20]
    b, n, ct, e_ch = 32, 80, 768, 256
    sd = {"proj.weight": torch.randn(ct, e_ch, generator=g) * 0.06, "proj.bias": torch.zeros(ct)}
    for l, ch in enumerate(chans):
        sd[f"projections.{l}.conv.weight"] = torch.randn(e_ch, ch, 1, 1, generator=g) * ch ** -0.5
        sd[f"projections.{l}.conv.bias"] = torch.zeros(e_ch)
    for nm, d in (("query", ct), ("key", e_ch), ("value", e_ch)):
        sd.update({f"{nm}.0.weight": torch.ones(d), f"{nm}.0.bias": torch.zeros(d),
                   f"{nm}.1.weight": torch.randn(e_ch, d, generator=g) * d ** -0.5, f"{nm}.1.bias": torch.zeros(e_ch)})
    feats = [torch.randn(b, ch, s, s, generator=g).to(dev).to(memory_format=torch.channels_last) for ch, s in zip(chans, sizes)]
    text = torch.randn(b, n, ct, generator=g).to(dev)
    for prec in ("fp32", "fp16x3"):
        m = ImagePoolingAttentionModule(chans, ct, e_ch, num_heads=8, precision=prec).load_state_dict(sd)
        us = timed(lambda: m(text, feats), reps=10)
        print(f"ImagePoolingAttentionModule [{prec}] B={b} N={n} levels 80/40/20: {us:8.1f} us")
    y = torch.randn(b * 80 * 80, e_ch, generator=g).to(dev)
    out = torch.empty(b * 9, e_ch, device=dev)
    us = timed(lambda: L.adaptive_maxpool_nhwc(y, out, 9 * e_ch, b, 80, 80, e_ch, 3))
    print(f"adaptive_maxpool_nhwc B={b} 80x80x{e_ch} -> 3x3: {us:8.1f} us  {y.numel() * 4 / us / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
