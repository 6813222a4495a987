"""Find which stage of a step is not hipGraph-capturable (each stage in its own process)."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import weights as W
from wedetect_amd.engine import ImageTower
from wedetect_amd.pack import pack
stage = sys.argv[1]
arch = sys.argv[2] if len(sys.argv) > 2 else "nano"
size = int(sys.argv[3]) if len(sys.argv) > 3 else 128
tower = ImageTower(arch, pack(W.make_state_dict(arch), arch), 1, size, size, max_classes=80)
x = torch.from_numpy(W.make_images(1, size, size)).cuda()
text = torch.from_numpy(W.make_text_bank(80)).cuda()
meta = tower.identity_meta()
def run():
    if stage == "backbone": tower.backbone(x)
    elif stage == "neck": tower.neck()
    elif stage == "head": tower.head()
    elif stage == "sim": tower.similarity(text, normalize=True)
    elif stage == "topk":
        from wedetect_amd import lib as L
        L.topk_candidates(tower.scores, 1, tower.ntot * 80, 0.001, tower.nms_pre, tower.cand_idx, tower.cand_score, tower.cand_count, tower.topk_ws)
    elif stage == "post": tower.postprocess(tower.scores.view(1, tower.ntot, 80), 0.001, meta, nms="mmcv")
    elif stage == "all": tower.detect(x, text, meta, normalize_text=True, score_thr=0.001, with_embed=True)
tower.detect(x, text, meta, normalize_text=True, score_thr=0.001, with_embed=True)
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print(stage, "capturing", flush=True)
with torch.cuda.graph(g):
    run()
print(stage, "captured", flush=True)
g.replay(); torch.cuda.synchronize()
print(stage, "replayed OK", flush=True)
# ---- mimic GraphedDetect usage: new inputs -> replay -> eager
from wedetect_amd.engine import GraphedDetect
if stage == "all":
    gd = GraphedDetect(tower, 80, normalize_text=True, score_thr=0.001)
    print("GraphedDetect built", flush=True)
    torch.cuda.synchronize()
    x1 = torch.from_numpy(W.make_images(1, size, size, seed=9)).cuda()
    out = gd(x1, text, meta)
    torch.cuda.synchronize()
    print("GraphedDetect replayed, count", out["count"].tolist(), flush=True)
    oe = tower.detect(x1, text, meta, normalize_text=True, score_thr=0.001, with_embed=True)
    torch.cuda.synchronize()
    print("eager after replay OK, count", oe["count"].tolist(), flush=True)
