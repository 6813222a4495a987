"""On-device A/B of the pre-split fp16x3 GEMM kernels (register-staged 50/51/55, direct-to-LDS 60) on the ConvNeXt MLP shapes of WeDetect-Base B=32 @640."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
torch.manual_seed(0)
reps = int(os.environ.get("REPS", "8"))
cfgs = [int(c) for c in os.environ.get("CFGS", "50,51,60").split(",")]
def timed(fn):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps
for name, m, n, k, act, res in (("s1_pw1", 819200, 512, 128, L.ACT_GELU, False), ("s1_pw2", 819200, 128, 512, L.ACT_NONE, True),
                               ("s2_pw1", 204800, 1024, 256, L.ACT_GELU, False), ("s2_pw2", 204800, 256, 1024, L.ACT_NONE, True),
                               ("s3_pw1", 51200, 2048, 512, L.ACT_GELU, False), ("s3_pw2", 51200, 512, 2048, L.ACT_NONE, True),
                               ("s4_pw1", 12800, 4096, 1024, L.ACT_GELU, False)):
    x = torch.randn(m, k, device="cuda"); g = torch.ones(k, device="cuda"); b0 = torch.zeros(k, device="cuda")
    xs = torch.empty_like(x); L.layernorm_rows(x, xs, g, b0, m, k, split=True)
    w = torch.randn(n, k, device="cuda") * k ** -0.5; bias = torch.randn(n, device="cuda")
    ws = L.split_weights(w)
    r = torch.randn(m, n, device="cuda") if res else None
    c = torch.empty(m, n, device="cuda")
    kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n, act=act)
    if res: kw.update(res=r, ldres=n)
    fl = L.SPLIT_A | (0 if res else L.SPLIT_C)
    out = [f"{name:7s}"]
    for cfg in cfgs:
        us = timed(lambda: L.conv_gemm(xs, None, bias, c, w_split=ws, split_cfg=cfg, split_flags=fl, **kw))
        out.append(f"cfg {cfg} {us:7.1f}us {2.0*m*n*k/us/1e6:6.1f}TF")
    print(" | ".join(out), flush=True)
