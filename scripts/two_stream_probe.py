"""Probe (round 5): does splitting the B = 32 step into two half batches on two HIP streams fill the chip better?
Every kernel of the step runs one workgroup per CU or fewer on the 40 x 40 / 20 x 20 maps and alternates between
MFMA-bound (GEMM) and HBM-bound (dwconv, LayerNorm) phases; two independent chains can overlap those.  Same kernels, same
results per image (the kernels are batch-invariant).  Prints images/s for: one tower B = 32; two towers B = 16 on two
streams; two towers B = 32 on two streams (two steps in flight)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wedetect_amd import weights as W
from wedetect_amd.engine import ImageTower
from wedetect_amd.pack import pack

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = pack(W.make_state_dict("base"), "base")
text = torch.from_numpy(W.make_text_bank(80)).cuda()
kw = dict(normalize_text=True, score_thr=0.001, with_embed=True)


def run(towers, imgs, streams, n):
    metas = [t.identity_meta() for t in towers]
    for m in metas:
        m[:, 7] = 1.0
    def step():
        for t, x, m, s in zip(towers, imgs, metas, streams):
            with torch.cuda.stream(s):
                t.detect(x, text, m, overlap_post=True, **kw)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def build(b, k):
    ts = [ImageTower("base", P, b, 640, 640, max_classes=80) for _ in range(k)]
    xs = [torch.from_numpy(W.make_images(b, 640, 640, seed=1234 + i)).cuda() for i in range(k)]
    for t, x in zip(ts, xs):
        t.calibrate(x)
    return ts, xs


main = torch.cuda.current_stream()
for rep in range(3):
    ts, xs = build(32, 1)
    dt = run(ts, xs, [main], steps)
    print(f"one tower  B=32, one stream : {32 / dt:8.1f} images/s  {1e3 * dt:.2f} ms per 32 images", flush=True)
    del ts, xs
    ts, xs = build(16, 2)
    s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s in s2:
        s.wait_stream(main)
    dt = run(ts, xs, s2, steps)
    print(f"two towers B=16, two streams: {32 / dt:8.1f} images/s  {1e3 * dt:.2f} ms per 32 images", flush=True)
    del ts, xs
    ts, xs = build(32, 2)
    dt = run(ts, xs, s2, steps)
    print(f"two towers B=32, two streams: {64 / dt:8.1f} images/s  {1e3 * dt / 2:.2f} ms per 32 images", flush=True)
    del ts, xs
    torch.cuda.empty_cache()
