"""Print the per-kernel-family GEMM table of one bench.py JSON line (stdin), largest share first."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], "images/s", d["ms_per_step"], "ms/step")
for k, v in sorted(d["gemm_kernels"].items(), key=lambda kv: -kv[1]["launches_per_step"] * kv[1]["avg_us"]):
    n, us = v["launches_per_step"], v["avg_us"]
    print(f"{k:45s} {n:3d} x {us:8.1f} us = {n * us / 1e3:6.2f} ms  {v['tflops']:6.1f} TF")
