"""dwconv7 + layernorm_rows (two kernels) against the fused wd_dwconv7_ln on the ConvNeXt-Base stage shapes, batch 32."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
dev = "cuda"
for name, (b, h, w, c) in {"s1 160x160x128": (32, 160, 160, 128), "tiny s1 160x160x96": (32, 160, 160, 96), "s2 80x80x256": (32, 80, 80, 256), "s3 40x40x512": (32, 40, 40, 512),
                           "s4 20x20x1024": (32, 20, 20, 1024)}.items():
    x = torch.randn(b * h * w, c, device=dev)
    w7 = torch.randn(49, c, device=dev) * 0.1
    bias, gam, bet = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    y = torch.empty_like(x)
    def pair():
        L.dwconv7(x, w7, bias, y, b, h, w, c)
        L.layernorm_rows(y, y, gam, bet, b * h * w, c, split=True)
    def fused():
        L.dwconv7_ln(x, w7, bias, y, gam, bet, b, h, w, c, split=True)
    res = {}
    for tag, fn in (("pair", pair), ("fused", fused), ("pair", pair), ("fused", fused)):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.setdefault(tag, []).append(1e3 * s.elapsed_time(e) / 10)
    nbytes = 2.0 * x.numel() * 4
    print(f"{name:16s} pair {min(res['pair']):8.1f} us ({nbytes / min(res['pair']) / 1e6:5.2f} TB/s)   fused {min(res['fused']):8.1f} us ({nbytes / min(res['fused']) / 1e6:5.2f} TB/s)", flush=True)
