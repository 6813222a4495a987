"""Small-batch latency of the whole hot path (device-resident inputs): eager launches vs the hipGraph replay the
detector classes use by default up to WEDETECT_GRAPH_MAX_BATCH images (engine.GraphedDetect), Base and Tiny at 640."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import weights as W
from wedetect_amd.engine import GraphedDetect, ImageTower
from wedetect_amd.pack import pack

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for arch in ("tiny", "base"):
    packed = pack(W.make_state_dict(arch, seed=1, num_prompts=0), arch)
    text = torch.from_numpy(W.make_text_bank(80)).cuda()
    for b in (1, 2, 4, 8):
        tower = ImageTower(arch, packed, b, 640, 640, max_classes=80)
        x = torch.from_numpy(W.make_images(b, 640, 640, seed=5)).cuda()
        meta = tower.identity_meta(); meta[:, 7] = 1.0
        kw = dict(normalize_text=True, score_thr=0.001, with_embed=False)
        res = {}
        for dag in (False, True):                      # round 5: the neck / head DAG on side streams, eager and captured
            tower.dag = tower._dag_in_capture = dag
            res[("eager", dag)] = timeit(lambda: tower.detect(x, text, meta, **kw))
            g = GraphedDetect(tower, 80, **kw)
            res[("graph", dag)] = timeit(lambda: g(x, text, meta))
            del g
        print(f"{arch:5s} batch {b}: eager {res[('eager', False)]:6.2f} ms  eager+DAG {res[('eager', True)]:6.2f} ms  hipGraph {res[('graph', False)]:6.2f} ms  "
              f"hipGraph+DAG {res[('graph', True)]:6.2f} ms   (best {b / min(res.values()) * 1e3:7.1f} images/s)", flush=True)
        del tower
