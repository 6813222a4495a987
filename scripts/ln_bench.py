"""wd_layernorm_rows_split on the ConvNeXt-Base stage shapes (batch 32): time per launch and HBM-side rate.  python scripts/ln_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from wedetect_amd import lib as L
for rows, c in ((204800, 256), (819200, 128), (51200, 512)):
    x = torch.randn(rows, c, device="cuda"); y = torch.empty_like(x)
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    for _ in range(3): L.layernorm_rows(x, y, g, b, rows, c, split=True)
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): L.layernorm_rows(x, y, g, b, rows, c, split=True)
        e.record(); torch.cuda.synchronize(); ts.append(1e3 * s.elapsed_time(e) / 10)
    print(f"R8={os.environ.get('WEDETECT_LN_R8','0')} rows {rows} c {c}: {min(ts):7.1f} us  {2*x.numel()*4/min(ts)/1e6:5.2f} TB/s  checksum {float(y.view(torch.int32).sum()):.0f}")
