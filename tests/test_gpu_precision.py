"""The two arithmetic modes against each other at the benchmark's size: WeDetect-Base, 640 x 640,
80-class normalised bank, whole step (tower -> similarity -> top-k -> NMS -> gather)."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, to_np

pytestmark = pytest.mark.gpu


def test_fp16x3_step_matches_fp32_step_at_full_size():
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    arch, b, hw, k = "base", 2, 640, 80
    packed = pack(W.make_state_dict(arch), arch)
    imgs = torch.from_numpy(W.make_images(b, hw, hw, seed=4321)).cuda()
    bank = torch.from_numpy(W.make_text_bank(k)).cuda()
    outs = {}
    for prec in ("fp32", "fp16x3"):
        tower = ImageTower(arch, packed, b, hw, hw, max_classes=k, precision=prec)
        meta = tower.identity_meta()
        meta[:, 7] = 1.0
        res = tower.detect(imgs, bank, meta, normalize_text=True, score_thr=0.001, with_embed=True)
        torch.cuda.synchronize()
        outs[prec] = dict(embed=tower.embed.clone(), scores=tower.scores.view(-1)[: b * tower.ntot * k].clone(),
                          boxes=tower.boxes.clone(), res={n: v.clone() for n, v in res.items()})
        del tower
    a, f = outs["fp32"], outs["fp16x3"]
    # the north-star tolerance is 1e-3; the split arithmetic stays two orders below it
    assert_close("region embeddings fp16x3 vs fp32", f["embed"], a["embed"], 5e-5, 1e-5)
    assert_close("scores fp16x3 vs fp32", f["scores"], a["scores"], 1e-5)
    assert_close("decoded boxes fp16x3 vs fp32", f["boxes"], a["boxes"], 1e-3)
    for i in range(b):
        na, nf = int(a["res"]["count"][i]), int(f["res"]["count"][i])
        assert na == nf == 300
        ka = set(zip(to_np(a["res"]["anchors"][i, :na]).tolist(), to_np(a["res"]["labels"][i, :na]).tolist()))
        kf = set(zip(to_np(f["res"]["anchors"][i, :nf]).tolist(), to_np(f["res"]["labels"][i, :nf]).tolist()))
        overlap = len(ka & kf) / 300.0
        assert overlap >= 0.97, f"image {i}: kept (anchor, class) sets overlap only {overlap:.3f}"
        assert_close("sorted kept scores", np.sort(to_np(f["res"]["scores"][i, :nf])), np.sort(to_np(a["res"]["scores"][i, :na])), 1e-5)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_step_is_bitwise_deterministic(precision):
    """No atomics or order-dependent reductions on the detect path: repeated steps give identical bits
    (direct-to-LDS / ping-pong kernels with hand-placed waits included)."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    arch, b, hw, k = "base", 4, 320, 80
    tower = ImageTower(arch, pack(W.make_state_dict(arch), arch), b, hw, hw, max_classes=k, precision=precision)
    imgs = torch.from_numpy(W.make_images(b, hw, hw, seed=99)).cuda()
    bank = torch.from_numpy(W.make_text_bank(k)).cuda()
    meta = tower.identity_meta()

    def run():
        r = tower.detect(imgs, bank, meta, normalize_text=True, score_thr=0.001, with_embed=True)
        torch.cuda.synchronize()
        return [tower.embed.clone(), tower.boxes.clone()] + [r[n].clone() for n in sorted(r)]

    first = run()
    for _ in range(6):
        assert all(torch.equal(x, y) for x, y in zip(run(), first))
