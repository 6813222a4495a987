"""What global -> LDS byte rate can a CU draw?  (wd_probe_lds_dma: 8 waves per workgroup, 8-16 one-KB LDS-DMA instructions in
flight per wave, no MFMA, no LDS reads.)  The LDS-fed fp16x3 GEMM kernels move 24-64 KB per K stage per workgroup; their
ceiling is this rate, not the MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L

dev = torch.device("cuda")
sink = torch.zeros(4096, dtype=torch.int32, device=dev)
buf = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
buf.random_(0, 255)


def run(window, grid, iters, pattern, pitch):
    fn = lambda: L.check(L.LIB.wd_probe_lds_dma(buf.data_ptr(), window, grid, iters, pattern, pitch, sink.data_ptr(), L.stream_ptr()), "probe")
    for _ in range(3):
        fn()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    return grid * 8 * iters * 8 * 1024 / t / 1e12, t * 1e6


print(f"{'working set / workgroup':>26s} {'WGs':>5s} {'pattern':>22s} {'TB/s':>7s} {'B/clk/CU @2.1GHz':>17s} {'us':>8s}")
for wgs in (256, 512):
    for window, tag in ((128 << 10, "128 KB (L2-resident)"), (512 << 10, "512 KB (MALL-resident)"), (4 << 20, "4 MB (> MALL: HBM)")):
        if wgs * window > buf.numel():
            continue
        for pattern, pitch, ptag in ((0, 0, "1 KB contiguous"), (1, 512, "16 rows x 64 B, pitch 512"), (1, 2048, "16 rows x 64 B, pitch 2048")):
            if pattern == 1 and 128 * pitch > window:
                continue
            iters = max(64, (64 << 20) // (8 * 8 * 1024) // 4)
            tb, us = run(window, wgs, iters, pattern, pitch)
            print(f"{tag:>26s} {wgs:5d} {ptag:>22s} {tb:7.2f} {tb * 1e12 / 256 / 2.1e9 * (256 / min(wgs, 256)):17.1f} {us:8.1f}")
