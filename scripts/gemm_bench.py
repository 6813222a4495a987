"""On-device A/B of wd_conv_gemm tile configurations (wd_conv_gemm_tuned) on the shapes
that dominate WeDetect-Base B=32 @640.  Prints TFLOP/s per (shape, config) and checks every
config against config 0's output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L

torch.manual_seed(0)
dev = "cuda"
SHAPES = {
    "s3_pw1 51200x2048x512 gelu": dict(m=51200, n=2048, k=512, act=L.ACT_GELU),
    "s3_pw1 51200x2048x512 noact": dict(m=51200, n=2048, k=512),
    "s3_pw2 51200x512x2048 res": dict(m=51200, n=512, k=2048, res=True),
    "s1_pw1 819200x512x128 gelu": dict(m=819200, n=512, k=128, act=L.ACT_GELU),
    "s2_pw2 204800x256x1024 res": dict(m=204800, n=256, k=1024, res=True),
    "plain 51200x128x1152 silu": dict(m=51200, n=128, k=1152, act=L.ACT_SILU),
    "plain 12800x256x2304 silu": dict(m=12800, n=256, k=2304, act=L.ACT_SILU),
    "s4_pw1 12800x4096x1024 gelu": dict(m=12800, n=4096, k=1024, act=L.ACT_GELU),
    "s4_pw2 12800x1024x4096 res": dict(m=12800, n=1024, k=4096, res=True),
    "sim 268800x80x768 sigm": dict(m=268800, n=80, k=768, sim=True),
    "conv3 128->128 @40 silu": dict(conv=(32, 40, 40, 128, 128)),
    "conv3 64->64 @80 silu": dict(conv=(32, 80, 80, 64, 64)),
    "conv3 256->256 @20 silu": dict(conv=(32, 20, 20, 256, 256)),
    "conv3 128->256 @80 silu": dict(conv=(32, 80, 80, 128, 256)),
    "conv3 512->256 @20 silu": dict(conv=(32, 20, 20, 512, 256)),
}
GROUPS = {"big": [3, 22, 13, 3, 22], "sim": [10, 19, 19], "conv": [3, 13]}
if os.environ.get("SIM_CFGS"):
    GROUPS["sim"] = [int(c) for c in os.environ["SIM_CFGS"].split(",")]
reps = int(os.environ.get("REPS", "8"))
only = os.environ.get("ONLY")
for name, sh in SHAPES.items():
    if only and only not in name:
        continue
    if "conv" in sh:
        bb, hh, ww, ci, co = sh["conv"]
        m, n, k = bb * hh * ww, co, 9 * ci
        a = torch.randn(bb * hh * ww, ci, device=dev)
    else:
        m, n, k = sh["m"], sh["n"], sh["k"]
        a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * k ** -0.5
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if sh.get("res") else None
    ref = None
    kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n, act=sh.get("act", L.ACT_NONE))
    if "conv" in sh:
        kw = dict(batch=bb, hin=hh, win=ww, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=n, ldc=n, act=L.ACT_SILU)
    if r is not None:
        kw.update(res=r, ldres=n)
    if sh.get("sim"):
        kw.update(sigmoid=True, seg=(8400, 6400, 8000, (0.7, 0.58, 0.82), (-2.6, -2.2, -1.9)))
    def run(cfg, c):
        if cfg is None:
            L.conv_gemm(a, w, b, c, **kw)
        else:
            L.conv_gemm(a, w, b, c, tuned_cfg=cfg, **kw)

    for cfg in GROUPS["sim" if sh.get("sim") else "conv" if "conv" in sh else "big"]:
        c = torch.empty(m, n, device=dev)
        try:
            run(cfg, c)
            torch.cuda.synchronize()
        except Exception as e:
            print(f"{name:32s} cfg {cfg!s:>3}: FAILED {e}")
            continue
        if ref is None:
            ref = c.clone()
            err = 0.0
        else:
            err = float((c - ref).abs().max())
        for _ in range(3):
            run(cfg, c)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            run(cfg, c)
        e.record()
        torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / reps
        print(f"{name:32s} cfg {cfg!s:>3}: {us:9.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF   max|d vs first| {err:.2e}", flush=True)
