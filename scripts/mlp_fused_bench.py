"""Stage-1 ConvNeXt block MLP: one fused launch (wd_mlp_fused_split) against the two-kernel chain, HIP-event timed.
    python scripts/mlp_fused_bench.py [rows]        (default 32 x 160 x 160 = WeDetect-Base batch 32 at 640)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L

m = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 160 * 160
c, h = 128, 512
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s, k=1.0: torch.randn(*s, device="cuda", generator=g) * k
x0, w1, b1, w2, b2 = r(m, c), r(h, c, k=c ** -0.5), r(h, k=0.1), r(c, h, k=h ** -0.5), r(c, k=0.1)
ws1, ws2 = L.split_weights(w1), L.split_weights(w2)
xs = torch.empty(m, c, device="cuda")
L.layernorm_rows(x0, xs, torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), m, c, split=True)
hid = torch.empty(m, h, device="cuda")
park = torch.zeros(L.p8_workspace_bytes() // 4, dtype=torch.float32, device="cuda")   # as the engine launches them


def chain(x):
    L.conv_gemm(xs, None, b1, hid, w_split=ws1, batch=1, hin=1, win=m, cin=c, lda=c, n=h, ldc=h, act=L.ACT_GELU,
                split_flags=L.SPLIT_A | L.SPLIT_C, workspace=park)
    L.conv_gemm(hid, None, b2, x, w_split=ws2, batch=1, hin=1, win=m, cin=h, lda=h, n=c, ldc=c, res=x, ldres=c,
                split_flags=L.SPLIT_A, workspace=park)


def fused(x):
    L.mlp_fused(xs, m, c, h, ws1, b1, ws2, b2, x)


def timeit(fn, n=20):
    x = x0.clone()
    for _ in range(3):
        fn(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


a = x0.clone()
chain(a)
flops = 3 * 2 * 2.0 * m * c * h
for rep in range(2):
    us = timeit(chain)
    print(f"two launches: {us:8.1f} us   {flops / us / 1e6:7.1f} TF (fp16x3 MFMA flops)")
    b = x0.clone()
    fused(b)
    torch.cuda.synchronize()
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    us = timeit(fused)
    print(f"fused       : {us:8.1f} us   {flops / us / 1e6:7.1f} TF   bit-identical to the two launches: {same}")
