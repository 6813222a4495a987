"""MFMA / VALU issue probe (wd_probe_issue): cycles per {1 MFMA + NV VALU} slot, MFMAs alone, VALU alone.
    PYTHONPATH=. python scripts/issue_probe.py"""
import ctypes
import torch
from wedetect_amd import lib as L

f = L.LIB.wd_probe_issue
out = torch.zeros(2, dtype=torch.int64, device="cuda")
KINDS = {0: "v_fma_f32", 1: "v_pk_fma_f32", 2: "v_exp_f32", 3: "v_mul_f32", 4: "v_cvt_pk_f16_f32", 5: "v_pk_mul_f32", 6: "v_pk_add_f32",
         7: "v_cmp+v_cndmask", 8: "v_rcp_f32", 9: "v_permlane32_swap", 10: "v_cvt_f32_f16", 11: "v_and_b32", 12: "v_cvt_f16_f32", 13: "v_fma 1 chain", 14: "v_fma 2 chains", 15: "v_fma 4 chains", 16: "v_fmaak 4 chains"}
ITERS = 4000


def run(mode, kind, nv, grid):
    sink = torch.zeros(grid * 256, device="cuda")
    for _ in range(2):
        L.check(f(mode, kind, nv, grid, ITERS, out.data_ptr(), sink.data_ptr(), L.stream_ptr()), "wd_probe_issue")
    torch.cuda.synchronize()
    t, r = [int(v) for v in out.tolist()]
    return t / (ITERS * 8), (t / (r / 100e6)) / 1e9 if r else 0.0     # ticks per slot, s_memtime ticks per second (GHz)


for grid, label in ((1, "one workgroup (1 wave/SIMD on one CU)"),):
    print(f"## {label}")
    m, ghz = run(1, 0, 0, grid)
    print(f"MFMA only                       : {m:7.1f} ticks/slot   ({ghz:.2f} Gticks/s)")
    md, _ = run(3, 0, 0, grid)
    print(f"MFMA only, one dependent chain  : {md:7.1f} ticks/slot")
    for nv in (4, 6, 8, 10):
        b, _ = run(4, 0, nv, grid)
        print(f"dependent MFMA + {nv:2d} x v_fma_f32 : both {b:7.1f}")
    for kind, nv in ((0, 2), (0, 4), (0, 6), (0, 7), (0, 8), (0, 10), (1, 4), (1, 7), (2, 2), (2, 4), (3, 7), (4, 4), (5, 4), (6, 4), (7, 3), (8, 2), (8, 4), (9, 4), (10, 6), (11, 6), (12, 6), (13, 3), (13, 6), (14, 6), (15, 6), (16, 6)):
        b, g1 = run(0, kind, nv, grid)
        v, g2 = run(2, kind, nv, grid)
        print(f"MFMA + {nv} x {KINDS[kind]:17s}: both {b:7.1f}   VALU alone {v:7.1f}   (sum {m + v:7.1f})   [{g1:.2f} / {g2:.2f} Gticks/s]")
