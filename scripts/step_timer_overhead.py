"""How much does per-launch kernel timing cost the step it measures?  Times the same Base B=32 step bare, with an
hipEventRecord pair around every wd_conv_gemm launch, and with every launch stamping a pre-created event pair from
its own dispatch (wd_time_next_gemm: what bench.py's GemmTimer does, on the last steps of its timed region only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L, weights as W
from wedetect_amd.engine import ImageTower
from wedetect_amd.pack import pack

B, S, K = 32, 640, 80
tower = ImageTower("base", pack(W.make_state_dict("base"), "base"), B, S, S, max_classes=K)
x = torch.from_numpy(W.make_images(B, S, S)).cuda()
text = torch.from_numpy(W.make_text_bank(K)).cuda()
meta = tower.identity_meta(); meta[:, 7] = 1.0
step = lambda: tower.detect(x, text, meta, normalize_text=True, score_thr=0.001, with_embed=True)

def timed(n=20):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

orig = L.conv_gemm
keep = []
def bracketed(*a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(*a, **k); e.record(); keep.append((s, e))
pool = []
def stamped(*a, **k):
    s, e = pool.pop()
    L.time_next_gemm(s, e); orig(*a, **k); keep.append((s, e))
for r in range(3):
    L.conv_gemm = orig
    bare = timed()
    L.conv_gemm = bracketed
    keep.clear()
    ev = timed()
    keep.clear()
    for _ in range(23 * 260):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); e.record(); pool.append((s, e))
    torch.cuda.synchronize()
    L.conv_gemm = stamped
    st = timed()
    pool.clear()
    print(f"bare {bare:.3f} ms/step ({B / bare * 1e3:.1f} img/s) | event pair recorded around every GEMM launch {ev:.3f} ms/step "
          f"({B / ev * 1e3:.1f} img/s) | every GEMM launch stamping its own events {st:.3f} ms/step ({B / st * 1e3:.1f} img/s)", flush=True)
