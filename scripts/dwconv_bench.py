"""wd_dwconv7 on the ConvNeXt-Base stage shapes (batch 32): time per launch and HBM-side rate (algorithmic bytes = one
read + one write of the activation), every kernel form that takes the shape (wd_dwconv7_variant: 2 = LDS tile / 1 x 4 strips,
3 = LDS tile / 1 x 8 strips; 0 = wd_dwconv7's own choice), with a
bit-identity check against form 2."""
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
    ref = None
    for variant in [int(v) for v in os.environ.get("VARIANTS", "2,3,4,0").split(",")]:
        if variant == 3 and h % 16:
            continue
        if variant in (2, 3, 4) and c % 32:
            continue
        y.zero_()
        for _ in range(3):
            L.dwconv7(x, w7, bias, y, b, h, w, c, variant=variant)
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                L.dwconv7(x, w7, bias, y, b, h, w, c, variant=variant)
            e.record()
            torch.cuda.synchronize()
            ts.append(1e3 * s.elapsed_time(e) / 10)
        if ref is None:
            ref = y.clone()
        same = "bit-identical" if torch.equal(ref, y) else f"DIFFERS max|d| {float((ref - y).abs().max()):.3e}"
        nbytes = 2.0 * x.numel() * 4
        print(f"{name:16s} form {variant}: {min(ts):8.1f} us ({nbytes / min(ts) / 1e6:5.2f} TB/s)  {same}", flush=True)
    if c % 32 == 0:                      # the LayerNorm-fold producer: hi/lo output + per-block statistics (wd_dwconv7_stats)
        ys = torch.empty_like(x)
        part = torch.empty(2 * b * h * w * (c // 32), device=dev)
        for _ in range(3):
            L.dwconv7_stats(x, w7, bias, ys, part, b, h, w, c)
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                L.dwconv7_stats(x, w7, bias, ys, part, b, h, w, c)
            e.record()
            torch.cuda.synchronize()
            ts.append(1e3 * s.elapsed_time(e) / 10)
        hv = ys.view(torch.float16).view(-1, c // 8, 2, 8)
        err = float(((hv[:, :, 0].float() + hv[:, :, 1].float()).reshape(-1, c) - ref).abs().max())
        print(f"{name:16s} stats  : {min(ts):8.1f} us ({2.0 * x.numel() * 4 / min(ts) / 1e6:5.2f} TB/s)  max|hi+lo - form 2| {err:.2e}", flush=True)
