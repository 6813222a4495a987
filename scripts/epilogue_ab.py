"""On-device A/B: what the GELU + fp16-split epilogue costs on the pwconv1 shapes of WeDetect-Base B=32 @640
(pre-split fp16x3 kernels 60 = direct-to-LDS 128x128, 63 = ping-pong 256x128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
torch.manual_seed(0)
reps = int(os.environ.get("REPS", "8"))
def timed(fn):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps
for name, m, n, k in (("s1_pw1", 819200, 512, 128), ("s2_pw1", 204800, 1024, 256), ("s3_pw1", 51200, 2048, 512)):
    x = torch.randn(m, k, device="cuda"); g = torch.ones(k, device="cuda"); b0 = torch.zeros(k, device="cuda")
    xs = torch.empty_like(x); L.layernorm_rows(x, xs, g, b0, m, k, split=True)
    w = torch.randn(n, k, device="cuda") * k ** -0.5; bias = torch.randn(n, device="cuda")
    ws = L.split_weights(w)
    c = torch.empty(m, n, device="cuda")
    for cfg in (60, 63):
        out = [f"{name} cfg {cfg}"]
        for label, act, fl in (("gelu+split", L.ACT_GELU, L.SPLIT_A | L.SPLIT_C), ("none+split", L.ACT_NONE, L.SPLIT_A | L.SPLIT_C),
                               ("relu+split", L.ACT_RELU, L.SPLIT_A | L.SPLIT_C), ("gelu fp32 out", L.ACT_GELU, L.SPLIT_A),
                               ("none fp32 out", L.ACT_NONE, L.SPLIT_A)):
            kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n, act=act)
            us = timed(lambda: L.conv_gemm(xs, None, bias, c, w_split=ws, split_cfg=cfg, split_flags=fl, **kw))
            out.append(f"{label} {us:7.1f}us")
        print(" | ".join(out), flush=True)
