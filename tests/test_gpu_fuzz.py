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
        a_split = buf.view(torch.float32).view(-1, k)[:m]          # the split buffer is padded to whole groups of 8 rows (round 5)
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


@pytest.mark.parametrize("seed", range(12))
def test_split_k_random_shapes(seed):
    """Forced and library-chosen K splits (partial sums in a workspace + ordered reduce with the full epilogue)
    on loader-split, pre-split staged and direct-to-LDS kernels, plain and conv, vs float64."""
    from wedetect_amd import lib as L
    g = np.random.default_rng(3000 + seed)
    m = int(g.integers(1, 500))
    n = int(g.integers(1, 40)) * 8
    k = int(g.integers(8, 80)) * 16
    act = int(g.integers(0, 4))
    use_res = bool(g.integers(0, 2))
    gen = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(m, k, device="cuda", generator=gen)
    w = torch.randn(n, k, device="cuda", generator=gen) * k ** -0.5
    b = torch.randn(n, device="cuda", generator=gen) * 0.3
    res = torch.randn(m, n, device="cuda", generator=gen) if use_res else None
    ref = _ref(a, w, b, act, res)
    tol = dict(atol=8e-6 * max(1.0, float(ref.abs().max())), rtol=2e-6)
    kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n, act=act)
    if use_res:
        kw.update(res=res, ldres=n)
    ws = L.split_weights(w)
    wsp = torch.empty(8 * m * n + 64, device="cuda")
    c = torch.empty(m, n, device="cuda")
    for splits in (0, 2, 3, 7):
        for cfg in (-1, 50, 51):
            c.fill_(float("nan"))
            L.conv_gemm(a, None, b, c, w_split=ws, split_cfg=cfg, workspace=wsp, k_splits=splits, **kw)
            assert_close(f"split-K {splits} cfg {cfg} m{m} n{n} k{k}", c, ref, **tol)
    buf = torch.empty(L.LIB.wd_split_weights_bytes(m, k), dtype=torch.uint8, device="cuda")
    L.check(L.LIB.wd_split_weights(a.data_ptr(), m, k, 1.0, buf.data_ptr(), L.stream_ptr()), "split a")
    a_split = buf.view(torch.float32).view(-1, k)[:m]          # the split buffer is padded to whole groups of 8 rows (round 5)
    for splits in (0, 2, 5):
        for cfg in (51, 60):
            c.fill_(float("nan"))
            L.conv_gemm(a_split, None, b, c, w_split=ws, split_cfg=cfg, split_flags=L.SPLIT_A, workspace=wsp, k_splits=splits, **kw)
            assert_close(f"pre-split split-K {splits} cfg {cfg}", c, ref, **tol)
    with pytest.raises(L.WedetectHipError):                     # workspace too small for a forced split
        L.conv_gemm(a, None, b, c, w_split=ws, workspace=torch.empty(16, device="cuda"), k_splits=2, **kw)


def test_split_k_conv_and_special_epilogue():
    from wedetect_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(5)
    bsz, h, w_, ci, co = 1, 10, 12, 64, 96
    x = torch.randn(bsz, h, w_, ci, device="cuda", generator=g)
    wt = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * (ci * 9) ** -0.5
    bias = torch.randn(co, device="cuda", generator=g)
    wrow = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), padding=1)
    ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1).reshape(-1, co)
    c = torch.empty(bsz * h * w_, co, device="cuda")
    wsp = torch.empty(6 * c.numel(), device="cuda")
    for splits in (0, 3, 6):
        c.fill_(float("nan"))
        L.conv_gemm(x, None, bias, c, batch=bsz, hin=h, win=w_, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=co, ldc=co,
                    act=L.ACT_SILU, w_split=L.split_weights(wrow), workspace=wsp, k_splits=splits)
        assert_close(f"conv split-K {splits}", c, ref, 8e-6 * max(1.0, float(ref.abs().max())), 2e-6)
    # per-level affine + sigmoid + batch-strided rows through the reduce kernel
    m, n, k = 2 * 84, 80, 768
    a = torch.randn(m, k, device="cuda", generator=g)
    w = torch.nn.functional.normalize(torch.randn(n, k, device="cuda", generator=g), dim=-1)
    seg = (84, 64, 80, (0.7, 0.58, 0.82), (-2.6, -2.2, -1.9))
    kw = dict(batch=2, hin=1, win=84, cin=k, lda=k, n=n, ldc=n, sigmoid=True, seg=seg)
    c0, c1 = torch.zeros(m, n, device="cuda"), torch.zeros(m, n, device="cuda")
    L.conv_gemm(a, w, None, c0, **kw)
    L.conv_gemm(a, None, None, c1, w_split=L.split_weights(w), workspace=torch.empty(4 * m * n, device="cuda"), k_splits=4, **kw)
    assert_close("split-K seg + sigmoid", c1, c0, 2e-6)


@pytest.mark.parametrize("seed", range(6))
def test_conv3_random_geometry_all_forms_bit_identical(seed):
    """Round 6: random 3 x 3 / stride 1 geometries on pre-split activations — maps narrower than a tile row, tiles that end mid-row
    and mid-image, ragged channel counts (cin % 16 == 0, n % 8 == 0), one- and two-stage K loops, residual and dual outputs, K
    splits — through the production choice (twelve-wave producer / consumer kernel), the eight-wave staggered form (cfg 77) and the
    in-step form of round 4 (cfg 78): the same bits, and within 2e-5 of a float64 convolution of the same (hi + lo) operands."""
    from wedetect_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(900 + seed)
    rnd = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g, device="cuda"))
    for _ in range(4):
        b_, h, w_ = rnd(1, 5), rnd(2, 45), rnd(2, 45)
        ci, co = 16 * rnd(1, 9), 8 * rnd(1, 40)
        m = b_ * h * w_
        x = torch.randn(m, ci, device="cuda", generator=g)
        wrow = torch.randn(co, 9 * ci, device="cuda", generator=g) * (9 * ci) ** -0.5
        bias = torch.randn(co, device="cuda", generator=g)
        res = torch.randn(m, co, device="cuda", generator=g)
        ws = L.split_weights(wrow)
        xs = torch.empty(m, ci, device="cuda")
        L.check(L.LIB.wd_split_weights(x.data_ptr(), m, ci, 1.0, xs.data_ptr(), L.stream_ptr()), "split")
        splits = rnd(1, 3)
        geo = dict(batch=b_, hin=h, win=w_, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=co, ldc=co, act=L.ACT_SILU, res=res, ldres=co,
                   res_alpha=0.25, w_split=ws)
        if splits > 1:
            geo.update(workspace=torch.empty(splits * m * co + 64, device="cuda"), k_splits=splits)
        outs = {}
        for cfg in (-1, 77, 78, 79):
            c = torch.full((m, co), float("nan"), device="cuda")
            c2 = torch.full((m, co), float("nan"), device="cuda")
            cs = torch.full((m, co), float("nan"), device="cuda")
            L.conv_gemm(xs, None, bias, c, split_flags=L.SPLIT_A, split_cfg=cfg, **geo)
            L.conv_gemm(xs, None, bias, cs, split_flags=L.SPLIT_A | L.SPLIT_C, split_cfg=cfg, c2=c2, ldc2=co, **geo)
            outs[cfg] = (c, cs, c2)
        what = f"b{b_} {h}x{w_} c{ci}->{co} splits {splits}"
        for cfg in (-1, 77, 79):
            for a, b in zip(outs[cfg], outs[78]):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"cfg {cfg} != cfg 78: {what}"
        assert torch.equal(outs[78][0], outs[78][2]), f"fp32 twin != fp32 output: {what}"
        # float64 reference on the split operands ([hi x8 | lo x8] groups)
        hl = xs.contiguous().view(torch.float16).view(m, ci // 8, 2, 8)
        xd = (hl[:, :, 0].double() + hl[:, :, 1].double()).reshape(m, ci)
        xd4 = xd.view(b_, h, w_, ci).permute(0, 3, 1, 2)
        wd = wrow.double().view(co, 3, 3, ci).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xd4, wd, bias.double(), padding=1).permute(0, 2, 3, 1).reshape(m, co)
        ref = torch.nn.functional.silu(ref) + 0.25 * res.double()
        err = float((outs[78][0].double() - ref).abs().max())
        assert err < 3e-5 * max(1.0, float(ref.abs().max())), f"{what}: max|d| {err:.3e}"
