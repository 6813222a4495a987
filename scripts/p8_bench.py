"""On-device A/B of the pre-split fp16x3 kernels on the ConvNeXt MLP shapes (Base B=32 @640): cfg 60 (128x128 direct
to LDS), 63 (256x128 ping-pong), 64 (256x256, four phases per K tile, counted DMA waits).  Operands are pre-split
(LayerNorm split output / C-split GELU output), random data; interleaved rounds, median reported.  TFLOP/s are
ALGORITHMIC (2 m n k / time); MFMA issue = 3x."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L

torch.manual_seed(0)
dev = "cuda"
SHAPES = {
    "s1_pw1 819200x512x128 gelu": dict(m=819200, n=512, k=128, gelu=True),
    "s1_pw2 819200x128x512 res": dict(m=819200, n=128, k=512, res=True),
    "s2_pw1 204800x1024x256 gelu": dict(m=204800, n=1024, k=256, gelu=True),
    "s2_pw2 204800x256x1024 res": dict(m=204800, n=256, k=1024, res=True),
    "s3_pw1 51200x2048x512 gelu": dict(m=51200, n=2048, k=512, gelu=True),
    "s3_pw2 51200x512x2048 res": dict(m=51200, n=512, k=2048, res=True),
    "e3_pw1 csplit noact": dict(m=51200, n=2048, k=512, csplit=True),
    "e3_pw1 fp32out gelu": dict(m=51200, n=2048, k=512, gelu=True, nosplit=True),
    "e3_pw1 fp32out noact": dict(m=51200, n=2048, k=512),
    "e1_pw1 csplit noact (stage 1)": dict(m=819200, n=512, k=128, csplit=True),
    "e1_pw1 fp32out gelu (stage 1)": dict(m=819200, n=512, k=128, gelu=True, nosplit=True),
    "e1_pw1 fp32out noact (stage 1)": dict(m=819200, n=512, k=128),
    "s4_pw1 12800x4096x1024 gelu": dict(m=12800, n=4096, k=1024, gelu=True),
    "s4_pw2 12800x1024x4096 res": dict(m=12800, n=1024, k=4096, res=True),
    # Base B = 8 (ONLY=b8): the stage 2 - 4 MLPs at a quarter of the rows
    "b8_s2_pw1 51200x1024x256 gelu": dict(m=51200, n=1024, k=256, gelu=True),
    "b8_s2_pw2 51200x256x1024 res": dict(m=51200, n=256, k=1024, res=True),
    "b8_s3_pw1 12800x2048x512 gelu": dict(m=12800, n=2048, k=512, gelu=True),
    "b8_s3_pw2 12800x512x2048 res": dict(m=12800, n=512, k=2048, res=True),
    "b8_s4_pw1 3200x4096x1024 gelu": dict(m=3200, n=4096, k=1024, gelu=True),
    "b8_s4_pw2 3200x1024x4096 res": dict(m=3200, n=1024, k=4096, res=True),
}
park = torch.zeros(L.p8_workspace_bytes() // 4, dtype=torch.float32, device=dev)
# cfg 60SS (e.g. 6002): cfg 60 with SS K splits through a workspace (fp32-output layers only)
kws = torch.empty(16 << 20, dtype=torch.float32, device="cuda")


def run(cfg, xs, b, c, ws, flags, kw):
    if cfg >= 6000:
        L.conv_gemm(xs, None, b, c, w_split=ws, split_cfg=60, split_flags=flags, workspace=kws, k_splits=cfg - 6000, **kw)
    else:
        L.conv_gemm(xs, None, b, c, w_split=ws, split_cfg=cfg, split_flags=flags, workspace=park if cfg == 65 else None, **kw)
cfgs = [int(c) for c in os.environ.get("CFGS", "60,63,64,65").split(",")]
rounds = int(os.environ.get("ROUNDS", "5"))
reps = int(os.environ.get("REPS", "6"))
only = os.environ.get("ONLY")
for name, sh in SHAPES.items():
    if only and only not in name:
        continue
    m, n, k = sh["m"], sh["n"], sh["k"]
    x = torch.randn(m, k, device=dev) * 1.5
    xs = torch.empty(L.LIB.wd_split_weights_bytes(m, k), dtype=torch.uint8, device=dev)      # [hi x8 | lo x8] groups, scale 1
    L.check(L.LIB.wd_split_weights(x.data_ptr(), m, k, 1.0, xs.data_ptr(), L.stream_ptr()), "wd_split_weights")
    xs = xs.view(torch.float32).view(-1, k)[:m]
    del x
    w = torch.randn(n, k, device=dev) * k ** -0.5
    b = torch.randn(n, device=dev)
    ws = L.split_weights(w)
    kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n)
    flags = L.SPLIT_A
    if sh.get("gelu"):
        kw.update(act=L.ACT_GELU)
    if (sh.get("gelu") and not sh.get("nosplit")) or sh.get("csplit"):
        flags |= L.SPLIT_C
    c = torch.empty(m, n, device=dev)
    if sh.get("res"):
        kw.update(res=c, ldres=n)                          # in place, as the engine runs pwconv2
    c_init = torch.randn(m, n, device=dev)
    times = {cfg: [] for cfg in cfgs}
    ok = {}
    ref = None
    for cfg in cfgs:
        try:
            c.copy_(c_init)
            run(cfg, xs, b, c, ws, flags, kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = c.clone()
                ok[cfg] = "ref"
            else:
                ok[cfg] = "bit-identical" if torch.equal(c.view(torch.int32), ref.view(torch.int32)) else f"DIFFERS max|d| {float((c - ref).abs().max()):.3e}"
        except Exception as ex:
            ok[cfg] = f"FAILED {ex}"
    for r in range(rounds):
        for cfg in cfgs:
            if ok[cfg].startswith("FAILED"):
                continue
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(cfg, xs, b, c, ws, flags, kw)
            s.record()
            for _ in range(reps):
                run(cfg, xs, b, c, ws, flags, kw)
            e.record()
            torch.cuda.synchronize()
            times[cfg].append(1e3 * s.elapsed_time(e) / reps)
    for cfg in cfgs:
        if not times[cfg]:
            print(f"{name:30s} cfg {cfg}: {ok[cfg]}", flush=True)
            continue
        med, best = statistics.median(times[cfg]), min(times[cfg])
        print(f"{name:30s} cfg {cfg}: median {med:8.1f} us {2.0*m*n*k/med/1e6:7.1f} TF  best {best:8.1f} us {2.0*m*n*k/best/1e6:7.1f} TF   {ok[cfg]}", flush=True)
    del xs, c, w, ws
    torch.cuda.empty_cache()
