"""ConvNeXt block MLP of the wide stages: one fused launch (wd_mlp_fused_wide, round 4) against the two-kernel chain (the
256 x 256 kernel twice), HIP-event timed.   python scripts/mlpw_bench.py [c] [rows]   (default c = 512, 32 x 40 x 40 rows =
WeDetect-Base batch 32 at 640, stage 3; c = 256: 32 x 80 x 80).  TFLOP/s are ALGORITHMIC (2 m n k per GEMM)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L

c = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = int(sys.argv[2]) if len(sys.argv) > 2 else (32 * 40 * 40 if c == 512 else 32 * 80 * 80)
h = 4 * c
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s, k=1.0: torch.randn(*s, device="cuda", generator=g) * k
x0, w1, b1, w2, b2 = r(m, c), r(h, c, k=c ** -0.5), r(h, k=0.1), r(c, h, k=h ** -0.5), r(c, k=0.1)
ws1, ws2 = L.split_weights(w1), L.split_weights(w2)
wf1, wf2 = (L.mlp_wide_pack(ws1[0], h, c), ws1[1]), (L.mlp_wide_pack(ws2[0], c, h), ws2[1])
xs = torch.empty(m, c, device="cuda")
L.layernorm_rows(x0, xs, torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), m, c, split=True)
hid = torch.empty(m, h, device="cuda")


def chain(x):
    L.conv_gemm(xs, None, b1, hid, w_split=ws1, batch=1, hin=1, win=m, cin=c, lda=c, n=h, ldc=h, act=L.ACT_GELU,
                split_flags=L.SPLIT_A | L.SPLIT_C)
    L.conv_gemm(hid, None, b2, x, w_split=ws2, batch=1, hin=1, win=m, cin=h, lda=h, n=c, ldc=c, res=x, ldres=c,
                split_flags=L.SPLIT_A)


park = torch.zeros(L.p8_workspace_bytes() // 4, dtype=torch.float32, device="cuda") if os.environ.get("PERSIST", "1") == "1" else None


def fused(x):
    L.mlp_fused_wide(xs, m, c, h, wf1, b1, wf2, b2, x, workspace=park)


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
flops = 2 * 2.0 * m * c * h
for rep in range(3):
    us = timeit(chain)
    print(f"c={c} m={m}  two launches: {us:8.1f} us   {flops / us / 1e6:7.1f} TF", flush=True)
    b = x0.clone()
    fused(b)
    torch.cuda.synchronize()
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    us = timeit(fused)
    print(f"c={c} m={m}  fused       : {us:8.1f} us   {flops / us / 1e6:7.1f} TF ({flops / us / 1e6 / 838.9:.3f} of 838.9)   bit-identical: {same}"
          + ("" if same else f"   max|d| {float((a - b).abs().max()):.3e}"), flush=True)
