"""In-kernel timeline of the staggered 3 x 3 kernels (timing-only build: python scripts/build_variant.py c3trace C3_TRACE, run with
WEDETECT_LIB=wedetect_amd/libwedetect_hip_c3trace.so): waves 0 and 4 (and the producer wave 8 of the twelve-wave form) of tile 0
stamp s_memtime on arrival at and release from every barrier.  Prints, per wave: prologue, per-phase work (release -> next
arrival) and barrier wait (arrival -> release) for a few stages, the loop's average per stage, the epilogue — in shader cycles."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
from conv_pp_bench import to_split

SHAPES = [(32, 40, 40, 128, 128), (32, 80, 80, 256, 256)]
CFGS = [int(c) for c in os.environ.get("CFGS", "77,79").split(",")]
g = torch.Generator(device="cuda").manual_seed(1)


def report_free(t, nk):
    """cfg 79: one barrier per stage — stamps: body start, then (arrive, release) per barrier: BARRIER(0), one per stage"""
    for wv in range(3):
        s = t[wv]
        nst = int((s[:4095] != 0).sum())
        st = s[:nst].tolist()
        end = int(s[4095])
        pairs = [(st[i], st[i + 1]) for i in range(1, nst - 1, 2)]
        per = [pairs[i + 1][1] - pairs[i][1] for i in range(len(pairs) - 1)]
        waits = [pairs[i][1] - pairs[i][0] for i in range(len(pairs))]
        print(f" wave {4 * wv}: {nst} stamps; prologue {pairs[0][1] - st[0]} cyc; release-to-release per stage: median {statistics.median(per):.0f} "
              f"(first {per[0]}, last {per[-1]}); barrier wait median {statistics.median(waits):.0f}; loop {pairs[-1][1] - pairs[0][1]} cyc; "
              f"after the last barrier {end - pairs[-1][1] if end else 0} cyc; whole {end - st[0] if end else 0} cyc")


def report(t, nk, cfg):
    if cfg == 79:
        return report_free(t, nk)
    for wv in range(2):
        s = t[wv]
        nst = int((s[:4095] != 0).sum())
        st = s[:nst].tolist()
        end = int(s[4095])
        pairs = [(st[i], st[i + 1]) for i in range(1, nst - 1, 2)]     # (arrive, release) per barrier; st[0] = body start
        extra = 1 if wv == 1 else 0                                    # group 1: one extra barrier before the loop
        pro = pairs[0][1] - st[0]
        loop0 = 1 + extra
        work_d = [pairs[i][0] - pairs[i - 1][1] for i in range(loop0, len(pairs))]
        wait_d = [pairs[i][1] - pairs[i][0] for i in range(loop0, len(pairs))]
        nph = 6 * nk
        loop = pairs[loop0 + nph - 1][1] - pairs[loop0 - 1][1]
        print(f" wave {4 * wv}: {nst} stamps; prologue {pro} cyc; loop {loop} cyc = {loop / nk:.0f} / stage; "
              f"epilogue {end - pairs[-1][1] if end else 0} cyc; whole {end - st[0] if end else 0} cyc")
        names = ["R0", "M0", "R1", "M1", "R2", "M2"] if wv < 2 else ["i0", "i1", "i2", "i3", "i4", "i5"]
        for stg in (0, 1, nk // 2, nk - 2, nk - 1):
            row = [f"{names[ph]} {work_d[stg * 6 + ph]:4d}+{wait_d[stg * 6 + ph]:4d}" for ph in range(6)]
            print(f"   stage {stg:3d}: " + "  ".join(row))
        for ph in range(6):
            ws_ = [work_d[stg * 6 + ph] for stg in range(2, nk - 2)]
            wt_ = [wait_d[stg * 6 + ph] for stg in range(2, nk - 2)]
            if ws_:
                print(f"   median {names[ph]}: work {statistics.median(ws_):.0f} wait {statistics.median(wt_):.0f}")


for (b, h, w, ci, n) in SHAPES:
    m = b * h * w
    x = torch.randn(m, ci, device="cuda", generator=g)
    wt = torch.randn(n, 9 * ci, device="cuda", generator=g) * (9 * ci) ** -0.5
    bias = torch.randn(n, device="cuda", generator=g)
    ws, xs = L.split_weights(wt), to_split(x)
    work = torch.zeros(3 * 4096 * 2 + 64, device="cuda", dtype=torch.float32)
    geo = dict(batch=b, hin=h, win=w, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=n, ldc=n, act=L.ACT_SILU, w_split=ws,
               split_flags=L.SPLIT_A | L.SPLIT_C, workspace=work, k_splits=1)
    c = torch.empty(m, n, device="cuda")
    nk = 3 * (ci // 16)
    for cfg in CFGS:
        for _ in range(3):
            L.conv_gemm(xs, None, bias, c, split_cfg=cfg, **geo)
        torch.cuda.synchronize()
        work.zero_()
        L.conv_gemm(xs, None, bias, c, split_cfg=cfg, **geo)
        torch.cuda.synchronize()
        t = work[: 3 * 4096 * 2].view(torch.int64).view(3, 4096).cpu()
        print(f"== {b}x{h}x{w} c{ci}->{n}: {nk} stages, cfg {cfg}")
        report(t, nk, cfg)
