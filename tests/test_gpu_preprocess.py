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
