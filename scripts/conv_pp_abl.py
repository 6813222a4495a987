"""Timing-only ablations of split_conv_pp_kernel (WD_DEBUG_ABLATIONS=1 build): which instruction stream bounds the K loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
from scripts.conv_pp_bench import to_split, timeit  # noqa

g = torch.Generator(device="cuda").manual_seed(1)
for (b, h, w, ci, n) in ((32, 40, 40, 128, 128), (32, 80, 80, 256, 256)):
    m = b * h * w
    x = torch.randn(m, ci, device="cuda", generator=g)
    wt = torch.randn(n, 9 * ci, device="cuda", generator=g) * (9 * ci) ** -0.5
    bias = torch.randn(n, device="cuda", generator=g)
    ws = L.split_weights(wt)
    xs = to_split(x)
    c = torch.empty(m, n, device="cuda")
    geo = dict(batch=b, hin=h, win=w, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=n, ldc=n, act=L.ACT_SILU, w_split=ws,
               split_flags=L.SPLIT_A | L.SPLIT_C)
    names = {74: "full (ring of 4)", 701: "no DMA", 702: "no ds_read", 703: "no DMA, no ds_read (MFMA + barriers)", 704: "no MFMA",
             707: "barriers only", 708: "no epilogue", 711: "MFMA only, no epilogue", 715: "barriers only, no epilogue"}
    print(f"{b}x{h}x{w} c{ci}->{n} 3x3")
    for cfg, nm in names.items():
        t = timeit(lambda: L.conv_gemm(xs, None, bias, c, split_cfg=cfg, **geo))
        print(f"   {nm:24s} {t:8.1f} us")
