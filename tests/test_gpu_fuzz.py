"""Randomised shape sweep over the dense kernels: every production path (fp32, fp16x3 loader-split,
pre-split register-staged / direct-to-LDS / ping-pong) on ragged m / n / k against float64 math."""
import numpy as np
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _ref(a, w, b, act, res):
    y = a.double() @ w.double().T + b.double()
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = y * torch.sigmoid(y)
    elif act == 3:
        y = torch.nn.functional.gelu(y)
    if res is not None:
        y = y + res.double()
    return y


@pytest.mark.parametrize("seed", range(24))
def test_plain_gemm_random_shapes_all_paths(seed):
    from wedetect_amd import lib as L
    g = np.random.default_rng(1000 + seed)
    m = int(g.integers(1, 900))
    n = int(g.integers(1, 60)) * 8
    k = int(g.integers(1, 40)) * 8
    act = int(g.integers(0, 4))
    use_res = bool(g.integers(0, 2))
    gen = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(m, k, device="cuda", generator=gen) * 1.5
    w = torch.randn(n, k, device="cuda", generator=gen) * k ** -0.5
    b = torch.randn(n, device="cuda", generator=gen) * 0.3
    res = torch.randn(m, n, device="cuda", generator=gen) if use_res else None
    ref = _ref(a, w, b, act, res)
    tol = dict(atol=8e-6 * max(1.0, float(ref.abs().max())), rtol=2e-6)
    kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n, act=act)
    if use_res:
        kw.update(res=res, ldres=n)
    c = torch.full((m, n), float("nan"), device="cuda")
    L.conv_gemm(a, w, b, c, **kw)
    assert_close(f"fp32 m{m} n{n} k{k}", c, ref, **tol)
    ws = L.split_weights(w)
    c.fill_(float("nan"))
    L.conv_gemm(a, None, b, c, w_split=ws, **kw)
    assert_close(f"fp16x3 loader-split m{m} n{n} k{k}", c, ref, **tol)
    # pre-split A: LayerNorm-free split of the same rows through wd_split_weights' layout (scale 1)
    a_split = torch.empty(m, k, device="cuda")
    buf = torch.empty(L.LIB.wd_split_weights_bytes(m, k), dtype=torch.uint8, device="cuda")
    if k % 16 == 0:                       # the weight splitter pads rows to 16: same bytes as [m, k] fp32 only then
        L.check(L.LIB.wd_split_weights(a.data_ptr(), m, k, 1.0, buf.data_ptr(), L.stream_ptr()), "split a")
        a_split = buf.view(torch.float32).view(m, k)
        for cfg in (-1, 51, 60, 63):
            c.fill_(float("nan"))
            L.conv_gemm(a_split, None, b, c, w_split=ws, split_cfg=cfg, split_flags=L.SPLIT_A, **kw)
            assert_close(f"fp16x3 pre-split cfg {cfg} m{m} n{n} k{k}", c, ref, **tol)


@pytest.mark.parametrize("seed", range(8))
def test_conv_random_geometry_fp16x3(seed):
    from wedetect_amd import lib as L
    g = np.random.default_rng(2000 + seed)
    bsz, h, w_ = int(g.integers(1, 4)), int(g.integers(3, 30)), int(g.integers(3, 30))
    ci, co = int(g.integers(1, 12)) * 8, int(g.integers(1, 30)) * 8
    kk, stride = ((3, 1), (3, 2), (2, 2))[int(g.integers(0, 3))]
    pad = 1 if kk == 3 else 0
    if kk == 2 and (h < 2 or w_ < 2):
        h, w_ = 4, 4
    gen = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(bsz, h, w_, ci, device="cuda", generator=gen)
    wt = torch.randn(co, ci, kk, kk, device="cuda", generator=gen) * (ci * kk * kk) ** -0.5
    bias = torch.randn(co, device="cuda", generator=gen)
    wrow = wt.permute(0, 2, 3, 1).reshape(co, kk * kk * ci).contiguous()
    ho, wo = (h + 2 * pad - kk) // stride + 1, (w_ + 2 * pad - kk) // stride + 1
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), stride=stride, padding=pad)
    ref = torch.relu(ref).permute(0, 2, 3, 1).reshape(-1, co)
    c = torch.full((bsz * ho * wo, co), float("nan"), device="cuda")
    L.conv_gemm(x, None, bias, c, batch=bsz, hin=h, win=w_, cin=ci, lda=ci, kh=kk, kw=kk, stride=stride, pad=pad, n=co,
                ldc=co, act=L.ACT_RELU, w_split=L.split_weights(wrow))
    assert_close(f"conv k{kk}s{stride} {bsz}x{h}x{w_}x{ci}->{co}", c, ref, 8e-6 * max(1.0, float(ref.abs().max())), 2e-6)
