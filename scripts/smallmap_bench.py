"""Neck / head 3x3 convs on the 40x40 and 20x20 maps (Base, batch 32): tile choices and split-K of wd_conv_gemm_split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
dev = "cuda"
B = int(os.environ.get("B", "32"))
SHAPES = {"rep_p4 40x40 128->128": (40, 40, 128, 128), "rep_n4 20x20 256->256": (20, 20, 256, 256), "rep_p3 80x80 64->64": (80, 80, 64, 64),
          "head cls 40x40 256->256": (40, 40, 256, 256), "head cls 20x20 512->256": (20, 20, 512, 256), "head reg 40x40 256->64": (40, 40, 256, 64)}
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for name, (h, w, ci, co) in SHAPES.items():
    m, n, k = B * h * w, co, 9 * ci
    a = torch.randn(m, ci, device=dev)
    wt = torch.randn(n, k, device=dev) * k ** -0.5
    b = torch.randn(n, device=dev)
    wsp = L.split_weights(wt)
    kw = dict(batch=B, hin=h, win=w, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=n, ldc=n, act=L.ACT_SILU)
    c = torch.empty(m, n, device=dev)
    ref = None
    for cfg, ks in ((-1, 0), (50, 1), (51, 1), (53, 1), (52, 1), (50, 2), (50, 3), (50, 4), (51, 2), (51, 4), (53, 2)):
        try:
            fn = (lambda: L.conv_gemm(a, None, b, c, w_split=wsp, split_cfg=cfg, workspace=ws, k_splits=ks, **kw)) if ks != 1 or cfg < 0 else \
                 (lambda: L.conv_gemm(a, None, b, c, w_split=wsp, split_cfg=cfg, **kw))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = c.clone()
            err = float((c - ref).abs().max())
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            us = 1e3 * s.elapsed_time(e) / 10
            print(f"{name:26s} m={m:6d} cfg {cfg:3d} ksplit {ks}: {us:7.1f} us {2.0*m*n*k/us/1e6:6.1f} TF  max|d| {err:.1e}", flush=True)
        except Exception as ex:
            print(f"{name:26s} cfg {cfg} ksplit {ks}: {str(ex)[:80]}", flush=True)
