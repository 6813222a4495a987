"""wd_dwconv7 on the ConvNeXt-Base stage shapes (batch 32): time per launch and HBM-side rate (algorithmic bytes = one
read + one write of the activation).  WEDETECT_DWCONV=0 selects the round-1 1 x 4-strip kernel for A/B (debug switch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
dev = "cuda"
for name, (b, h, w, c) in {"s1 160x160x128": (32, 160, 160, 128), "s2 80x80x256": (32, 80, 80, 256), "s3 40x40x512": (32, 40, 40, 512),
                           "s4 20x20x1024": (32, 20, 20, 1024), "odd 37x53x96": (3, 37, 53, 96)}.items():
    x = torch.randn(b * h * w, c, device=dev)
    w7 = torch.randn(49, c, device=dev) * 0.1
    bias = torch.randn(c, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        L.dwconv7(x, w7, bias, y, b, h, w, c)
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            L.dwconv7(x, w7, bias, y, b, h, w, c)
        e.record()
        torch.cuda.synchronize()
        ts.append(1e3 * s.elapsed_time(e) / 10)
    nbytes = 2.0 * x.numel() * 4
    print(f"{name:16s} {min(ts):8.1f} us ({nbytes / min(ts) / 1e6:5.2f} TB/s)  checksum {float(y.double().sum()):.6f} {float(y.double().abs().sum()):.6f}", flush=True)
