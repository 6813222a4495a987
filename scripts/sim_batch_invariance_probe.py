"""Is the fp16x3 similarity path batch-invariant?  Same image alone and as image 0 of a batch of two, same split scales, then
with the embedding split scale halved: max |score difference|, kept-list equality."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import weights as W
from wedetect_amd.engine import ImageTower
from wedetect_amd.pack import pack
arch, hw = "nano", 128
P = pack(W.make_state_dict(arch, num_prompts=256), arch)
t2, t1 = ImageTower(arch, P, 2, hw, hw), ImageTower(arch, P, 1, hw, hw)
imgs = torch.from_numpy(W.make_images(2, hw, hw)).cuda()
t2.calibrate(imgs)
print("scales", {k: v for k, v in t2.sscale.items() if k in ("embed",)}, len(t2.sscale))
for label, sc in (("same scales", dict(t2.sscale)), ("embed scale / 2", dict(t2.sscale, embed=t2.sscale.get("embed", 1.0) / 2)),
                  ("own calibration", None)):
    if sc is None:
        t1.calibrate(imgs[:1])
        print("  t1 scales that differ:", {k: (t1.sscale.get(k), t2.sscale.get(k)) for k in set(t1.sscale) | set(t2.sscale) if t1.sscale.get(k) != t2.sscale.get(k)})
    else:
        t1.adopt_scales(sc)
    for mode in ("auto", "0"):
        t1.sim_split = t2.sim_split = mode
        e2, _ = t2.features(imgs, num_classes=256)
        s2 = t2.similarity(t2.P["prompts"], normalize=False).clone()
        e1, _ = t1.features(imgs[:1], num_classes=256)
        s1 = t1.similarity(t1.P["prompts"], normalize=False).clone()
        torch.cuda.synchronize()
        print(f"{label:18s} sim_split={mode}: embed equal {torch.equal(e2[0], e1[0])}  scores max|d| {float((s2[0] - s1[0]).abs().max()):.3e}  equal {torch.equal(s2[0], s1[0])}")
