"""Workload for rocprofv3 passes over the 3 x 3 kernels: each form (cfg 78 in-step, 77 staggered, 79 / 80 twelve waves) runs the two
benchmark layers 20 times; the kernel names tell the forms apart (template arguments), $SHAPE picks the layer (0 / 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
from conv_pp_bench import to_split
SHAPES = [(32, 40, 40, 128, 128), (32, 80, 80, 256, 256)]
b, h, w, ci, n = SHAPES[int(os.environ.get("SHAPE", "0"))]
g = torch.Generator(device="cuda").manual_seed(1)
m = b * h * w
x = torch.randn(m, ci, device="cuda", generator=g)
wt = torch.randn(n, 9 * ci, device="cuda", generator=g) * (9 * ci) ** -0.5
bias = torch.randn(n, device="cuda", generator=g)
ws, xs = L.split_weights(wt), to_split(x)
c = torch.empty(m, n, device="cuda")
geo = dict(batch=b, hin=h, win=w, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=n, ldc=n, act=L.ACT_SILU, w_split=ws, split_flags=L.SPLIT_A | L.SPLIT_C)
for cfg in [int(c_) for c_ in os.environ.get("CFGS", "78,77,79,80").split(",")]:
    for _ in range(20):
        L.conv_gemm(xs, None, bias, c, split_cfg=cfg, **geo)
    torch.cuda.synchronize()
