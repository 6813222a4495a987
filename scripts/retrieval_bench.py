"""TFLOP/s of wd_retrieval_max on the BASELINE configs[4] shape: 300 regions x 1M-class bank."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
n_img, rows, dim = 8, 300, 768
for k in (125_000, 1_000_000):
    e = torch.randn(n_img, rows, dim, device="cuda") * 1.4
    t = torch.nn.functional.normalize(torch.randn(k, dim, device="cuda"), dim=-1)
    scale = torch.full((n_img, rows), -0.35, device="cuda"); bias = torch.full((n_img, rows), -2.6, device="cuda")
    cnt = torch.full((n_img,), 300, dtype=torch.int32, device="cuda")
    out = torch.empty(n_img, k, device="cuda")
    for _ in range(2):
        L.retrieval_max(e, t, scale, bias, cnt, out, n_img, rows, k, dim)
    s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    reps = 3
    for _ in range(reps):
        L.retrieval_max(e, t, scale, bias, cnt, out, n_img, rows, k, dim)
    f.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(f) / reps
    fl = 2.0 * n_img * rows * dim * k
    print(f"retrieval_max {n_img} img x {rows} regions x {k} classes: {ms:.2f} ms  {fl/ms/1e9:.1f} TF  "
          f"({n_img/ms*1e3:.1f} img/s; bank read {k*dim*4/1e9:.2f} GB x {n_img} images)", flush=True)
    ts = L.split_weights(t)
    out2 = torch.empty(n_img, k, device="cuda")
    for _ in range(2):
        L.retrieval_max_split(e, ts, scale, bias, cnt, out2, n_img, rows, k, dim)
    s.record()
    for _ in range(reps):
        L.retrieval_max_split(e, ts, scale, bias, cnt, out2, n_img, rows, k, dim)
    f.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(f) / reps
    print(f"retrieval_max_split (fp16x3) same problem: {ms:.2f} ms  {fl/ms/1e9:.1f} TF  ({n_img/ms*1e3:.1f} img/s)  "
          f"max|d vs fp32| {float((out2 - out).abs().max()):.2e}", flush=True)
