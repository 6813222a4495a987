"""Device letterbox (wd_letterbox_u8) parity: bit-exact against the reference's letterbox goldens,
against PIL on larger images, and through the detector surface."""
import numpy as np
import pytest
import torch

from tests.util import golden, to_np

pytestmark = pytest.mark.gpu


def test_device_letterbox_reproduces_reference_goldens():
    from wedetect_amd.preprocess import DeviceLetterbox
    fx = golden("letterbox.npz")
    for i in range(int(fx["count"])):
        th, tw, ratio, dw, dh = fx[f"meta{i}"]
        lb = DeviceLetterbox((int(th), int(tw)))
        out, ratios, pads = lb([fx[f"img{i}"]])
        assert np.array_equal(to_np(out[0]), fx[f"out{i}"]), f"golden {i}"
        assert ratios[0] == ratio and pads[0] == (dw, dh)


def test_device_letterbox_matches_pil_on_a_ragged_batch():
    """One call, images of different sizes (down- and up-scaling, tall / wide / tiny), PIL inputs,
    numpy inputs and device tensors mixed: every canvas equals the host letterbox bit for bit."""
    from PIL import Image
    from wedetect_amd.detector import letterbox
    from wedetect_amd.preprocess import DeviceLetterbox
    g = np.random.default_rng(21)
    sizes = [(1080, 1920), (375, 500), (2000, 1500), (64, 48), (640, 640), (17, 5), (333, 777), (9, 1201)]
    arrs = [g.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    batch = [Image.fromarray(arrs[0]), arrs[1], torch.from_numpy(arrs[2]).cuda(), torch.from_numpy(arrs[3])] + arrs[4:]
    lb = DeviceLetterbox((640, 640))
    out, ratios, pads = lb(batch)
    assert out.shape == (len(sizes), 640, 640, 3) and out.dtype == torch.uint8
    for i, a in enumerate(arrs):
        ref, r, pad = letterbox(Image.fromarray(a), (640, 640))
        assert np.array_equal(to_np(out[i]), np.asarray(ref)), f"image {i} {sizes[i]}"
        assert ratios[i] == r and pads[i] == pad
    # a second call reuses the cached tables and the scratch buffer
    out2, _, _ = lb(batch)
    assert torch.equal(out, out2)


def test_letterbox_rejects_bad_inputs():
    from wedetect_amd import lib as L
    from wedetect_amd.preprocess import DeviceLetterbox
    lb = DeviceLetterbox((64, 64))
    with pytest.raises(L.WedetectHipError):
        lb([np.zeros((8, 8), np.uint8)])
    with pytest.raises(L.WedetectHipError):
        lb([np.zeros((8, 8, 3), np.float32)])
    with pytest.raises(L.WedetectHipError):
        lb([np.zeros((8, 8, 3), np.uint8)], out=torch.empty(1, 32, 32, 3, dtype=torch.uint8, device="cuda"))


def test_device_test_pipeline_geometry_and_pixels():
    """DeviceTestPipeline: metadata = the reference transforms' (fixture), padding is 114 outside the pasted rectangle,
    and the pixels inside equal the PIL-exact letterbox kernel's for the same target rectangle."""
    import json
    import os
    from tests.util import GOLDEN
    from wedetect_amd.preprocess import DeviceLetterbox, DeviceTestPipeline, letterbox_geometry, mmdet_test_geometry
    recs = {(r["h"], r["w"]): r for r in json.load(open(os.path.join(GOLDEN, "mmdet_geometry.json"))) if r["scale"] == [640, 640]}
    g = np.random.default_rng(4)
    sizes = [(720, 1280), (480, 640), (333, 500), (32, 32), (640, 640), (1080, 1920)]
    imgs = [g.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    canvas, metas = DeviceTestPipeline((640, 640))(imgs)
    assert tuple(canvas.shape) == (len(sizes), 640, 640, 3) and canvas.dtype == torch.uint8
    ref_canvas, _, _ = DeviceLetterbox((640, 640))(imgs)
    same = 0
    for i, (h, w) in enumerate(sizes):
        r, m = recs[(h, w)], metas[i]
        assert list(m["scale_factor"]) == r["scale_factor"] and m["pad_param"].tolist() == r["pad_param"]
        assert m["ori_shape"] == (h, w) and tuple(m["img_shape"]) == (640, 640)
        geo = mmdet_test_geometry(h, w, (640, 640))
        nh, nw = geo["no_pad_shape"]
        top, left = int(geo["pad_param"][0]), int(geo["pad_param"][2])
        c = canvas[i].cpu().numpy()
        mask = np.ones((640, 640), bool)
        mask[top:top + nh, left:left + nw] = False
        assert np.all(c[mask] == 114)
        lw, lh, lleft, ltop, _, _ = letterbox_geometry(w, h, (640, 640))
        if (lh, lw, ltop, lleft) == (nh, nw, top, left):   # same rectangle as the PIL letterbox -> same pixels
            assert np.array_equal(c, ref_canvas[i].cpu().numpy()), (h, w)
            same += 1
    assert same >= 4
