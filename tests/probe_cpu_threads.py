"""Probe tool (not collected by pytest): time the CPU oracle (Base@640, one image) at several thread counts on this host.
    python -m tests.probe_cpu_threads   (output kept in profiles/r01_cpu_threads_probe.txt)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as orc, postprocess as opp
from wedetect_amd import weights as W
from wedetect_amd.arch import get_arch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
a = get_arch("base"); sd = orc.to_torch(W.make_state_dict("base")); text = torch.from_numpy(W.make_text_bank(80))
imgs = W.make_images(1, 640, 640)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    ts = []
    for rep in range(2):
        t0 = time.perf_counter()
        with torch.no_grad():
            _, p = orc.forward_features(sd, a, imgs)
            t1 = time.perf_counter()
            flat = orc.head_flat(sd, p, text, normalize_text=True)
            t2 = time.perf_counter()
        opp.mmdet_predict_image(flat["boxes"][0].numpy(), flat["scores"][0].numpy(), None, (1.0, 1.0), (640, 640))
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    print(nt, "threads: net %.2fs head %.2fs post %.2fs" % ts[-1], flush=True)
