"""What the 256 x 256 fp16x3 K loop sustains on this chip with NOTHING else in the launch: the p8 kernel without its epilogue
(-DWD_DEBUG_ABLATIONS build, cfg 640 + 4; wrong results by construction, timing only) on random data, stage-3 shapes of
WeDetect-Base at batch 32, back-to-back launches for ~0.4 s per arm with the shader clock sampled from sysfs every 10 ms.
This is the `power_limited_ceiling` of bench.py's roofline: the fraction of the fp16x3 roof (2516.6 / 3 TF at 2.4 GHz) a launch
of this instruction mix can reach when its epilogue is free.  Needs WEDETECT_LIB = an ablation build
(python scripts/build_variant.py abl WD_DEBUG_ABLATIONS).  Prints one JSON line per arm."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
from bench import ClockSampler

ROOF = 2516.6 / 3
torch.manual_seed(0)
park = torch.zeros(L.p8_workspace_bytes() // 4, dtype=torch.float32, device="cuda")
ARMS = [("pw1 51200x2048x512", 51200, 2048, 512), ("pw2 51200x512x2048", 51200, 512, 2048), ("sq 51200x2048x2048", 51200, 2048, 2048)]
for name, m, n, k in ARMS:
    x = torch.randn(m, k, device="cuda") * 1.5
    xs = torch.empty(L.LIB.wd_split_weights_bytes(m, k), dtype=torch.uint8, device="cuda")
    L.check(L.LIB.wd_split_weights(x.data_ptr(), m, k, 1.0, xs.data_ptr(), L.stream_ptr()), "split")
    xs = xs.view(torch.float32).view(-1, k)[:m]
    w = torch.randn(n, k, device="cuda") * k ** -0.5
    ws = L.split_weights(w)
    c = torch.zeros(m, n, device="cuda")
    kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n)
    for label, cfg, work in (("k-loop only (no epilogue)", 644, None), ("tile form, fp32 rows out", 64, None), ("persistent form", 65, park)):
        def launch():
            L.conv_gemm(xs, None, None, c, w_split=ws, split_cfg=cfg, split_flags=L.SPLIT_A, workspace=work, **kw)
        try:
            for _ in range(5):
                launch()
            torch.cuda.synchronize()
        except Exception as ex:
            print(json.dumps({"arm": name, "kernel": label, "error": str(ex)[:120]}), flush=True)
            continue
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); launch(); e.record(); torch.cuda.synchronize()
        reps = max(20, int(0.4e3 / max(s.elapsed_time(e), 1e-3)))
        clk = ClockSampler(torch.cuda.current_device())
        clk.start()
        s.record()
        for _ in range(reps):
            launch()
        e.record()
        torch.cuda.synchronize()
        c_ = clk.stop()
        us = 1e3 * s.elapsed_time(e) / reps
        tf = 2.0 * m * n * k / us / 1e6
        mhz = c_["mean_mhz"] if c_ else None
        print(json.dumps({"arm": name, "kernel": label, "launches": reps, "avg_us": round(us, 1), "tflops": round(tf, 1),
                          "frac_of_fp16x3_roof": round(tf / ROOF, 4), "effective_mhz": mhz,
                          "frac_at_effective_clock": round(tf / ROOF * 2400.0 / mhz, 4) if mhz else None}), flush=True)
    del x, xs, w, ws, c
    torch.cuda.empty_cache()
