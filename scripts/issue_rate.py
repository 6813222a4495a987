"""Host-side issue time of a pipelined step against its device time (is a small batch bound by the host's launch rate?):
python scripts/issue_rate.py [batch ...]"""
import sys, time
import torch
from wedetect_amd import weights as W
from wedetect_amd.engine import ImageTower
from wedetect_amd.pack import pack

for B in [int(a) for a in sys.argv[1:]] or [1, 2, 8, 32]:
    t = ImageTower("base", pack(W.make_state_dict("base"), "base"), B, 640, 640, max_classes=80)
    x = torch.from_numpy(W.make_images(B, 640, 640, seed=1)).cuda()
    text = torch.from_numpy(W.make_text_bank(80)).cuda()
    meta = t.identity_meta(); meta[:, 7] = 1.0
    kw = dict(normalize_text=True, score_thr=0.001, with_embed=True, overlap_post=True)
    t.calibrate(x)
    for _ in range(8):
        t.detect(x, text, meta, **kw)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        t.detect(x, text, meta, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: host issue {1e3 * (t1 - t0) / n:.3f} ms / step, wall {1e3 * (t2 - t0) / n:.3f} ms / step, depth {t._bb_depth()}", flush=True)
    del t
