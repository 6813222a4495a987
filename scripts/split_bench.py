"""On-device A/B of the fp16x3 GEMM (wd_conv_gemm_split tile configurations) against the fp32
MFMA kernel on the shapes that dominate WeDetect-Base B=32 @640.  TFLOP/s are ALGORITHMIC
(2 m n k / time); the fp16x3 kernel issues three MFMA passes per product.  Every result is also
checked against a float64 matmul on a sample of rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L

torch.manual_seed(0)
dev = "cuda"
SHAPES = {
    "s1_pw1 819200x512x128 gelu": dict(m=819200, n=512, k=128, act=L.ACT_GELU),
    "s1_pw2 819200x128x512 res": dict(m=819200, n=128, k=512, res=True),
    "s2_pw1 204800x1024x256 gelu": dict(m=204800, n=1024, k=256, act=L.ACT_GELU),
    "s2_pw2 204800x256x1024 res": dict(m=204800, n=256, k=1024, res=True),
    "s3_pw1 51200x2048x512 gelu": dict(m=51200, n=2048, k=512, act=L.ACT_GELU),
    "s3_pw1 51200x2048x512 noact": dict(m=51200, n=2048, k=512),
    "s3_pw2 51200x512x2048 res": dict(m=51200, n=512, k=2048, res=True),
    "s4_pw1 12800x4096x1024 gelu": dict(m=12800, n=4096, k=1024, act=L.ACT_GELU),
    "s4_pw2 12800x1024x4096 res": dict(m=12800, n=1024, k=4096, res=True),
    "conv3 128->256 @80 silu": dict(conv=(32, 80, 80, 128, 256)),
    "conv3 256->256 @80 silu": dict(conv=(32, 80, 80, 256, 256)),
    "conv3 128->128 @40 silu": dict(conv=(32, 40, 40, 128, 128)),
    "conv3 256->256 @20 silu": dict(conv=(32, 20, 20, 256, 256)),
    "small conv3 512->256 @20 silu": dict(conv=(32, 20, 20, 512, 256)),
    "small conv3 256->128 @40 silu": dict(conv=(32, 40, 40, 256, 128)),
    "small plain 12800x256x768 silu": dict(m=12800, n=256, k=768, act=L.ACT_SILU),
    "small plain 51200x128x384 silu": dict(m=51200, n=128, k=384, act=L.ACT_SILU),
    "n64 conv3 64->64 @80 silu": dict(conv=(32, 80, 80, 64, 64)),
    "n64 conv3 128->64 @80 silu": dict(conv=(32, 80, 80, 128, 64)),
    "n64 plain 204800x64x128 silu": dict(m=204800, n=64, k=128, act=L.ACT_SILU),
    "n64 conv3 64->64 @40 silu": dict(conv=(32, 40, 40, 64, 64)),
}
cfgs = [int(c) for c in os.environ.get("CFGS", "0,8,1,9,2,10,3,11,4,12,5,13").split(",")]
reps = int(os.environ.get("REPS", "8"))
only = os.environ.get("ONLY")
for name, sh in SHAPES.items():
    if only and only not in name:
        continue
    if "conv" in sh:
        bb, hh, ww, ci, co = sh["conv"]
        m, n, k = bb * hh * ww, co, 9 * ci
        a = torch.randn(m, ci, device=dev)
        kw = dict(batch=bb, hin=hh, win=ww, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=n, ldc=n, act=L.ACT_SILU)
    else:
        m, n, k = sh["m"], sh["n"], sh["k"]
        a = torch.randn(m, k, device=dev)
        kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n, act=sh.get("act", L.ACT_NONE))
    w = torch.randn(n, k, device=dev) * k ** -0.5
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if sh.get("res") else None
    if r is not None:
        kw.update(res=r, ldres=n)
    ws = L.split_weights(w)
    ref = torch.empty(m, n, device=dev)
    L.conv_gemm(a, w, b, ref, **kw)
    torch.cuda.synchronize()

    def timed(fn):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return 1e3 * s.elapsed_time(e) / reps

    c = torch.empty(m, n, device=dev)
    us = timed(lambda: L.conv_gemm(a, w, b, c, **kw))
    print(f"{name:30s} fp32 {L.gemm_config(m, n, k):>14s}: {us:8.1f} us {2.0*m*n*k/us/1e6:7.1f} TF", flush=True)
    for cfg in ([-1, 6, 7, 14, 15, 16, 17, 9, 10] if n == 64 and "CFGS" not in os.environ else cfgs):
        c.zero_()
        try:
            L.conv_gemm(a, None, b, c, w_split=ws, split_cfg=cfg, **kw)
            torch.cuda.synchronize()
        except Exception as ex:
            print(f"{name:30s} split cfg {cfg:2d}: FAILED {ex}")
            continue
        err = float((c - ref).abs().max())
        us = timed(lambda: L.conv_gemm(a, None, b, c, w_split=ws, split_cfg=cfg, **kw))
        print(f"{name:30s} split cfg {cfg:2d}: {us:8.1f} us {2.0*m*n*k/us/1e6:7.1f} TF   max|d vs fp32| {err:.2e}", flush=True)
