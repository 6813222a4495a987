"""Kernel-level parity tests (GPU): every C-ABI entry point against a CPU reference on the
same seeded inputs.  Floating-point kernels: fp64/fp32 torch-CPU references, tolerance
written per test.  Index kernels (top-k, NMS): bit-exact against oracle/postprocess.py
and the committed golden fixtures."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, golden, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from wedetect_amd import lib
    return lib


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("m,n,k", [(1, 48, 48), (100, 64, 64), (300, 80, 768), (257, 81, 100), (1000, 96, 256),
                                   (513, 128, 512), (130, 256, 36), (77, 1203, 768), (4096, 512, 2048),
                                   (300, 192, 32), (64, 16, 4)])
def test_gemm_plain(L, m, n, k):
    a, w, b = rnd(1, m, k), rnd(2, n, k, scale=k ** -0.5), rnd(3, n)
    c = torch.full((m, n), 7.0, device="cuda")
    L.conv_gemm(dev(a), dev(w), dev(b), c, batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    assert_close(f"gemm {m}x{n}x{k} [{L.gemm_config(m, n, k)}]", c, ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("act", ["none", "relu", "silu", "gelu"])
def test_gemm_activation_residual_slice(L, act):
    m, n, k, ldc, off = 333, 64, 128, 200, 24
    a, w, b, r = rnd(4, m, k), rnd(5, n, k, scale=k ** -0.5), rnd(6, n), rnd(7, m, n)
    full = torch.full((m, ldc), -3.0, device="cuda")
    code = dict(none=L.ACT_NONE, relu=L.ACT_RELU, silu=L.ACT_SILU, gelu=L.ACT_GELU)[act]
    L.conv_gemm(dev(a), dev(w), dev(b), full[:, off:], batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=ldc, act=code,
                res=dev(r), ldres=n, res_alpha=0.625)
    v = torch.from_numpy(a.astype(np.float64) @ w.astype(np.float64).T + b)
    v = dict(none=lambda x: x, relu=F.relu, silu=F.silu, gelu=F.gelu)[act](v)
    ref = v.numpy() + 0.625 * r
    out = to_np(full)
    assert_close(f"gemm+{act}+res slice", out[:, off:off + n], ref, atol=3e-5, rtol=3e-5)
    assert np.all(out[:, :off] == -3.0) and np.all(out[:, off + n:] == -3.0), "wrote outside its channel slice"


def test_gemm_inplace_residual(L):
    m, n, k = 640, 128, 512
    a, w, b, x = rnd(8, m, k), rnd(9, n, k, scale=k ** -0.5), rnd(10, n), rnd(11, m, n)
    xd = dev(x)
    L.conv_gemm(dev(a), dev(w), dev(b), xd, batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n, res=xd, ldres=n)
    assert_close("x += a w^T + b (in place)", xd, a.astype(np.float64) @ w.astype(np.float64).T + b + x, 3e-5, 3e-5)


def test_gemm_similarity_epilogue(L):
    """Per-level exp(logit_scale)/bias + sigmoid over rows grouped per image (seg mode), unaligned N."""
    b_, ntot, k, kcls = 3, 84, 768, 81
    e, t = rnd(12, b_ * ntot, k), rnd(13, kcls, k)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    sc, bs = (0.7, 0.58, 0.82), (-2.6, -2.2, -1.9)
    out = torch.empty(b_, ntot, kcls, device="cuda")
    L.conv_gemm(dev(e), dev(t), None, out, batch=1, hin=1, win=b_ * ntot, cin=k, lda=k, n=kcls, ldc=kcls,
                sigmoid=True, seg=(ntot, 64, 80, sc, bs))
    lvl = (np.arange(ntot) >= 64).astype(int) + (np.arange(ntot) >= 80).astype(int)
    lg = (e.astype(np.float64) @ t.astype(np.float64).T).reshape(b_, ntot, kcls)
    lg = lg * np.asarray(sc)[lvl][None, :, None] + np.asarray(bs)[lvl][None, :, None]
    assert_close("sim sigmoid seg", out, 1 / (1 + np.exp(-lg)), atol=2e-6, rtol=1e-5)


def test_gemm_batch_stride_rows(L):
    """A level's rows land at b*stride + pos inside a taller [B, Ntot, C] tensor."""
    b_, hw, ntot, off, k, n = 3, 16, 21, 4, 64, 96
    a, w = rnd(14, b_ * hw, k), rnd(15, n, k)
    big = torch.full((b_, ntot, n), 9.0, device="cuda")
    L.conv_gemm(dev(a), dev(w), None, big.view(-1, n)[off:], batch=b_, hin=4, win=4, cin=k, lda=k, n=n, ldc=n,
                c_batch_stride=ntot)
    ref = (a.astype(np.float64) @ w.astype(np.float64).T).reshape(b_, hw, n)
    out = to_np(big)
    assert_close("batch-stride rows", out[:, off:off + hw], ref, 2e-5, 2e-5)
    assert np.all(out[:, :off] == 9.0) and np.all(out[:, off + hw:] == 9.0)


@pytest.mark.parametrize("cin,cout,k,stride,h,w", [(32, 64, 3, 1, 9, 7), (48, 48, 3, 1, 16, 16), (64, 96, 3, 2, 16, 12),
                                                   (128, 256, 2, 2, 8, 8), (96, 80, 1, 1, 5, 6), (256, 256, 3, 1, 20, 20),
                                                   (64, 64, 3, 2, 7, 9)])
def test_conv_nhwc(L, cin, cout, k, stride, h, w):
    b_ = 2
    pad = 1 if k == 3 else 0
    x, wt, bias = rnd(16, b_, cin, h, w), rnd(17, cout, cin, k, k, scale=(cin * k * k) ** -0.5), rnd(18, cout)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(bias).double(),
                   stride=stride, padding=pad)
    ho, wo = ref.shape[2:]
    lda = cin + 8                                   # input lives in a channel slice of a wider buffer
    xin = torch.zeros(b_, h, w, lda, device="cuda")
    xin[..., 4:4 + cin] = dev(np.transpose(x, (0, 2, 3, 1)))
    wrows = np.transpose(wt, (0, 2, 3, 1)).reshape(cout, -1)
    out = torch.empty(b_ * ho * wo, cout, device="cuda")
    L.conv_gemm(xin[..., 4:], dev(wrows), dev(bias), out, batch=b_, hin=h, win=w, cin=cin, lda=lda, kh=k, kw=k,
                stride=stride, pad=pad, n=cout, ldc=cout)
    assert_close(f"conv{k}x{k}s{stride} {cin}->{cout} @{h}x{w}", out.view(b_, ho, wo, cout),
                 ref.permute(0, 2, 3, 1).numpy(), 3e-5, 3e-5)


def test_deconv2x2(L):
    b_, ci, co, h, w = 2, 64, 32, 5, 6
    x, wt, bias = rnd(19, b_, ci, h, w), rnd(20, ci, co, 2, 2, scale=ci ** -0.5), rnd(21, co)
    ref = F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                             torch.from_numpy(bias).double(), stride=2)
    ldc = 3 * co
    out = torch.full((b_, 2 * h, 2 * w, ldc), 5.0, device="cuda")
    wrows = np.transpose(wt, (2, 3, 1, 0)).reshape(4 * co, ci)
    L.conv_gemm(dev(np.transpose(x, (0, 2, 3, 1))), dev(wrows), dev(np.tile(bias, 4)), out, batch=b_, hin=h, win=w,
                cin=ci, lda=ci, n=4 * co, ldc=ldc, out_mode=L.OUT_DECONV2X2)
    o = to_np(out)
    assert_close("deconv2x2", o[..., :co], ref.permute(0, 2, 3, 1).numpy(), 3e-5, 3e-5)
    assert np.all(o[..., co:] == 5.0)


def test_gemm_rejects_bad_args(L):
    a = torch.zeros(8, 6, device="cuda")
    with pytest.raises(L.WedetectHipError):
        L.conv_gemm(a, a, None, a, batch=1, hin=1, win=8, cin=6, lda=6, n=8, ldc=8)       # cin % 4 != 0


# ------------------------------------------------------------------------------------------ elementwise
def test_stem_patchify(L):
    img = np.random.default_rng(22).integers(0, 256, size=(2, 32, 24, 3), dtype=np.uint8)
    out = torch.empty(2 * 8 * 6, 48, device="cuda")
    L.stem_patchify(torch.from_numpy(img).cuda(), out)
    x = torch.from_numpy(img).float() / 255.0                                        # [B,H,W,3]
    ref = x.view(2, 8, 4, 6, 4, 3).permute(0, 1, 3, 2, 4, 5).reshape(2 * 8 * 6, 48)
    assert torch.equal(out.cpu(), ref), "patchify must be bit-exact (uint8 -> float, one division)"


@pytest.mark.parametrize("c0,b,h,w", [(128, 2, 64, 48), (96, 1, 36, 44), (192, 3, 20, 28), (64, 1, 8, 12), (128, 32, 160, 160),
                                      (128, 5, 640, 644)])
def test_stem_fused_bit_identical_to_the_three_kernels(L, c0, b, h, w):
    """wd_stem_fused (patchify + 4 x 4 stride-4 conv + bias + LayerNorm in one kernel, the weight matrix in registers as
    MFMA fragments) against wd_stem_patchify -> wd_conv_gemm (fp32 kernel) -> wd_layernorm_rows: the same K order per
    accumulator, the same reduction trees, the same bits.  Widths 64 / 96 / 128 / 192 (LayerNorm groups of 16, 32, 32, 64
    lanes), pixel counts that are not a multiple of 16, map widths that are not, rows >= 65536 (the four-rows-per-group
    LayerNorm) and fewer; and against float64 math."""
    rng = np.random.default_rng(31)
    img = torch.from_numpy(rng.integers(0, 256, size=(b, h, w, 3), dtype=np.uint8)).cuda()
    wt, bias = dev(rnd(32, c0, 48, scale=0.3)), dev(rnd(33, c0, scale=0.1))
    g, bt = dev(rnd(34, c0) * 0.2 + 1.0), dev(rnd(35, c0) * 0.1)
    m = b * (h // 4) * (w // 4)
    patches = torch.empty(m, 48, device="cuda")
    want = torch.empty(m, c0, device="cuda")
    L.stem_patchify(img, patches)
    L.conv_gemm(patches, wt, bias, want, batch=b, hin=h // 4, win=w // 4, cin=48, lda=48, n=c0, ldc=c0)
    L.layernorm_rows(want, want, g, bt, m, c0)
    got = torch.full((m, c0), float("nan"), device="cuda")
    L.stem_fused(img, wt, bias, g, bt, got)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), f"max|d| {float((got - want).abs().max())}"
    if m <= 4096:
        x = (img.double() / 255.0).view(b, h // 4, 4, w // 4, 4, 3).permute(0, 1, 3, 2, 4, 5).reshape(m, 48)
        ref = F.layer_norm(x @ wt.double().T + bias.double(), (c0,), g.double(), bt.double(), 1e-6)
        assert_close(f"stem c0={c0}", got, ref.cpu().numpy(), 2e-5, 2e-5)


def test_stem_fused_rejects_bad_arguments(L):
    img = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device="cuda")
    z = lambda *s_: torch.zeros(*s_, device="cuda")
    with pytest.raises(L.WedetectHipError):
        L.stem_fused(img, z(80, 48), z(80), z(80), z(80), z(4, 80))              # width without an instantiation
    with pytest.raises(L.WedetectHipError):
        L.stem_fused(torch.zeros(1, 8, 6, 3, dtype=torch.uint8, device="cuda"), z(128, 48), z(128), z(128), z(128), z(2, 128))


@pytest.mark.parametrize("c,h,w", [(32, 16, 16), (96, 9, 11), (128, 20, 20), (192, 5, 3), (512, 8, 8), (1536, 2, 2), (64, 64, 48), (32, 80, 37),
                                   (128, 160, 160)])
def test_dwconv7(L, c, h, w):
    b_ = 2
    x, wt, bias = rnd(23, b_, c, h, w), rnd(24, c, 1, 7, 7, scale=1 / 7), rnd(25, c)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(bias).double(),
                   padding=3, groups=c)
    y = torch.empty(b_, h, w, c, device="cuda")
    L.dwconv7(dev(np.transpose(x, (0, 2, 3, 1))), dev(wt.reshape(c, 49).T), dev(bias), y, b_, h, w, c)
    assert_close(f"dwconv7 c{c} {h}x{w}", y, ref.permute(0, 2, 3, 1).numpy(), 1e-5, 1e-5)


@pytest.mark.parametrize("c,h,w", [(32, 16, 16), (128, 20, 20), (512, 40, 40), (64, 64, 48), (32, 80, 37), (96, 9, 11), (1024, 7, 21), (256, 33, 18)])
def test_dwconv7_kernel_forms_are_bit_identical(L, c, h, w):
    """The generic kernel (form 1), the 1 x 4-strip tile kernel (2), the 1 x 8-strip tile kernel (3, h % 16 == 0) and the
    LDS-DMA staged tile kernel (4, round 6) accumulate every output in the same order (bias, taps kh-major, kw ascending):
    identical bits, ragged tiles and map borders included; unsupported forms are refused (wd_dwconv7_variant)."""
    b_ = 3
    x, wt, bias = rnd(61, b_, h, w, c), rnd(62, 49, c, scale=1 / 7), rnd(63, c)
    xd, wd_, bd = dev(x), dev(wt), dev(bias)
    outs = {}
    for variant in (1, 2, 3, 4, 0):
        if variant == 3 and h % 16:
            with pytest.raises(L.WedetectHipError):
                L.dwconv7(xd, wd_, bd, torch.empty(b_, h, w, c, device="cuda"), b_, h, w, c, variant=variant)
            continue
        y = torch.full((b_, h, w, c), float("nan"), device="cuda")
        L.dwconv7(xd, wd_, bd, y, b_, h, w, c, variant=variant)
        outs[variant] = y
    for variant, y in outs.items():
        assert torch.equal(y, outs[1]), f"form {variant} differs from the generic kernel: max|d| {float((y - outs[1]).abs().max()):.3e}"
    ref = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt.T.reshape(c, 1, 7, 7).copy()).double(),
                   torch.from_numpy(bias).double(), padding=3, groups=c)
    assert_close(f"dwconv7 forms c{c} {h}x{w}", outs[2], ref.permute(0, 2, 3, 1).numpy(), 1e-5, 1e-5)
    with pytest.raises(L.WedetectHipError):
        L.dwconv7(xd[..., :24].contiguous(), wd_[:, :24].contiguous(), bd[:24], torch.empty(b_, h, w, 24, device="cuda"), b_, h, w, 24, variant=2)


@pytest.mark.parametrize("c", [32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536])
def test_layernorm_rows(L, c):
    rows = 77
    x, g, b = rnd(26, rows, c, scale=2.0) + 0.5, rnd(27, c) * 0.2 + 1.0, rnd(28, c) * 0.1
    ref = F.layer_norm(torch.from_numpy(x), (c,), torch.from_numpy(g), torch.from_numpy(b), 1e-6)
    xd = dev(x)
    y = torch.empty_like(xd)
    L.layernorm_rows(xd, y, dev(g), dev(b), rows, c)
    assert_close(f"layernorm c{c}", y, ref, 5e-6, 5e-6)
    L.layernorm_rows(xd, xd, dev(g), dev(b), rows, c)        # in place
    assert torch.equal(xd, y)


def test_l2norm_rows(L):
    x = rnd(29, 81, 768) * 3
    y = torch.empty(81, 768, device="cuda")
    L.l2norm_rows(dev(x), y)
    assert_close("l2norm", y, F.normalize(torch.from_numpy(x), dim=-1), 1e-6, 1e-6)


def test_dfl_decode(L):
    from oracle import ref_cpu as orc
    b_, sizes = 2, [(8, 6), (4, 3), (2, 2)]
    ntot = sum(h * w for h, w in sizes)
    boxes = torch.zeros(b_, ntot, 4, device="cuda")
    pri, stride = orc.grid_priors(sizes)
    off, ref = 0, []
    for l, (h, w) in enumerate(sizes):
        d = rnd(30 + l, b_, h * w, 64, scale=2.0)
        L.dfl_decode(dev(d), 64, boxes, b_, h, w, (8, 16, 32)[l], off, ntot)
        dd = torch.from_numpy(d).view(b_, h * w, 4, 16).softmax(3).matmul(torch.arange(16.).view(-1, 1)).squeeze(-1)
        dd = dd * (8, 16, 32)[l]
        p = pri[off:off + h * w][None]
        ref.append(torch.stack([p[..., 0] - dd[..., 0], p[..., 1] - dd[..., 1], p[..., 0] + dd[..., 2],
                                p[..., 1] + dd[..., 3]], -1))
        off += h * w
    assert_close("dfl decode", boxes, torch.cat(ref, 1), 2e-4, 1e-6)


# ------------------------------------------------------------------------------------------ top-k
def run_topk(L, scores_list, thr, nms_pre):
    b_ = len(scores_list)
    n = scores_list[0].size
    s = dev(np.stack([x.reshape(-1) for x in scores_list]))
    cap = L.topk_capacity(nms_pre)
    ws = torch.empty(L.topk_workspace_bytes(b_, n, nms_pre) + 256, dtype=torch.uint8, device="cuda")
    o = (-ws.data_ptr()) % 256
    ws = ws[o:]
    idx = torch.empty(b_, cap, dtype=torch.int32, device="cuda")
    sc = torch.empty(b_, cap, device="cuda")
    cnt = torch.empty(b_, dtype=torch.int32, device="cuda")
    L.topk_candidates(s, b_, n, thr, nms_pre, idx, sc, cnt, ws)
    torch.cuda.synchronize()
    return to_np(idx), to_np(sc), to_np(cnt)


def check_topk(L, scores_list, thr, nms_pre, name):
    from oracle import postprocess as opp
    idx, sc, cnt = run_topk(L, scores_list, np.float32(thr), nms_pre)
    for i, s in enumerate(scores_list):
        rs, rl, ra = opp.filter_scores_and_topk(s, thr, nms_pre)
        k = s.shape[1]
        assert cnt[i] == rs.shape[0], f"{name}[{i}]: count {cnt[i]} vs oracle {rs.shape[0]}"
        n = int(cnt[i])
        ref_flat = ra * k + rl
        bad = np.nonzero(idx[i, :n] != ref_flat)[0]
        assert bad.size == 0, (f"{name}[{i}]: {bad.size}/{n} candidate indices differ, first at rank {bad[0]}: "
                               f"got {idx[i, bad[0]]} (score {sc[i, bad[0]]}) want {ref_flat[bad[0]]} (score {rs[bad[0]]})")
        assert np.array_equal(sc[i, :n], rs), f"{name}[{i}]: candidate scores differ"
        assert np.all(idx[i, n:] == -1)


def test_topk_against_oracle(L):
    g = np.random.default_rng(31)
    a = g.random((700, 9), dtype=np.float32)
    ties = (np.round(g.random((700, 9), dtype=np.float32) * 16) / 16).astype(np.float32)
    low = (g.random((700, 9), dtype=np.float32) * 1e-4).astype(np.float32)
    const = np.full((700, 9), 0.5, dtype=np.float32)
    check_topk(L, [a, ties, low, const], 0.001, 1000, "mixed batch, truncating")
    check_topk(L, [a, ties, low, const], 0.001, 30000, "mixed batch, everything kept")
    check_topk(L, [ties], 0.3, 50, "ties at the cut")
    check_topk(L, [low], 0.5, 100, "empty")


def test_topk_large_sigmoid_map(L):
    g = np.random.default_rng(32)
    maps = [(1 / (1 + np.exp(-(g.standard_normal((8400, 256)) - 2.5)))).astype(np.float32) for _ in range(2)]
    check_topk(L, maps, 0.0, 30000, "8400x256 sigmoid")


def test_topk_golden(L):
    fx = golden("filter_topk.npz")
    for name in ("ties", "trunc", "empty", "all_equal"):
        s = fx[f"{name}.in"]
        idx, sc, cnt = run_topk(L, [s], np.float32(fx[f"{name}.thr"]), int(fx[f"{name}.topk"]))
        n = int(cnt[0])
        assert n == fx[f"{name}.scores"].shape[0], name
        assert np.array_equal(idx[0, :n], fx[f"{name}.anchors"] * s.shape[1] + fx[f"{name}.labels"]), name
        assert np.array_equal(sc[0, :n], fx[f"{name}.scores"]), name


# ------------------------------------------------------------------------------------------ NMS
def run_nms(L, boxes_list, scores_list, labels_list, k, thr, max_out, meta_rows, embed=None, mode="vanilla", param=0):
    """Candidates are fed pre-sorted, one candidate per anchor (flat = anchor*k + label)."""
    mode = {"vanilla": L.NMS_VANILLA, "torchvision": L.NMS_TORCHVISION, "mmcv": L.NMS_MMCV}[mode]
    b_ = len(boxes_list)
    nmax = max(1, max(b.shape[0] for b in boxes_list))
    boxes = np.zeros((b_, nmax, 4), np.float32)
    cidx = np.full((b_, nmax), -1, np.int32)
    csc = np.zeros((b_, nmax), np.float32)
    cnt = np.zeros(b_, np.int32)
    for i, (bx, sc, lb) in enumerate(zip(boxes_list, scores_list, labels_list)):
        n = bx.shape[0]
        boxes[i, :n] = bx
        cidx[i, :n] = np.arange(n) * k + lb
        csc[i, :n] = sc
        cnt[i] = n
    ob = torch.empty(b_, max_out, 4, device="cuda")
    os_ = torch.empty(b_, max_out, device="cuda")
    ol = torch.empty(b_, max_out, dtype=torch.int32, device="cuda")
    oa = torch.empty(b_, max_out, dtype=torch.int32, device="cuda")
    oc = torch.empty(b_, dtype=torch.int32, device="cuda")
    oe = None
    ed = None
    if embed is not None:
        ed = dev(embed)
        oe = torch.empty(b_, max_out, embed.shape[-1], device="cuda")
    ws = torch.full((L.nms_workspace_bytes(b_) // 4,), 0x5a5a5a5a, dtype=torch.int32, device="cuda")   # the call must clear it
    L.nms_gather(dev(cidx, torch.int32), dev(csc), dev(cnt, torch.int32), nmax, dev(boxes), nmax, k,
                 dev(np.asarray(meta_rows, np.float32)), L.nms_threshold(thr, mode), max_out, ed,
                 0 if embed is None else embed.shape[-1], ob, os_, ol, oa, oc, oe, b_, nms_mode=mode, mode_param=param,
                 workspace=ws)
    torch.cuda.synchronize()
    return to_np(ob), to_np(os_), to_np(ol), to_np(oa), to_np(oc), None if oe is None else to_np(oe)


IDENT = [0, 0, 0, 1, 1, 1e9, 1e9, 0]


def test_nms_golden_and_random(L):
    from oracle import postprocess as opp
    fx = golden("nms.npz")
    cases = [(fx["unit.boxes"], fx["unit.scores"], fx["unit.labels"], fx["unit.keep"]),
             (fx["rand.boxes"], fx["rand.scores"], fx["rand.labels"], fx["rand.keep"]),
             (np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64), fx["empty.keep"])]
    max_out = 300
    ob, os_, ol, oa, oc, _ = run_nms(L, [c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], 7, 0.7,
                                     max_out, [IDENT] * 3)
    for i, (bx, sc, lb, keep) in enumerate(cases):
        keep = keep[:max_out]
        assert oc[i] == keep.shape[0], f"nms case {i}: kept {oc[i]} vs oracle {keep.shape[0]}"
        assert np.array_equal(oa[i, :oc[i]], keep), f"nms case {i}: kept indices differ"
        assert np.array_equal(ol[i, :oc[i]], lb[keep])
        assert np.array_equal(os_[i, :oc[i]], sc[keep])
        assert oc[i] == 0 or np.array_equal(ob[i, :oc[i]], opp.clamp_boxes(bx[keep], (1e9, 1e9)))
        assert np.all(oa[i, oc[i]:] == -1)
    # dense overlaps, many classes, more than max_out survivors, both rescale orders
    g = np.random.default_rng(33)
    n = 5000
    ctr = g.random((n, 2), dtype=np.float32) * 300 + 20
    wh = g.random((n, 2), dtype=np.float32) * 80 + 2
    bx = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    sc = np.sort(g.random(n, dtype=np.float32))[::-1].copy()
    lb = g.integers(0, 3, n).astype(np.int64)
    emb = g.standard_normal((1, n, 32)).astype(np.float32)
    meta_pre = [6.0, 4.0, 0, 0.5, 0.8, 500.0, 380.0, 1.0]
    meta_post = [6.0, 4.0, 0, 0.5, 0.8, 500.0, 380.0, 0.0]
    for meta, split_thr in ((meta_pre, 10000), (meta_pre, 1000), (meta_post, 0)):
        if meta[7]:     # mmdet order: rescale, then mmcv.ops.batched_nms (one agnostic call / per-class loop), clamp
            ob, os_, ol, oa, oc, oe = run_nms(L, [bx], [sc], [lb], 3, 0.7, 300, [meta], embed=emb, mode="mmcv", param=split_thr)
            r = opp.mmdet_predict_image_from_candidates(bx, sc, lb, (meta[1], 0, meta[0], 0), (meta[3], meta[4]),
                                                        (meta[6], meta[5]), 0.7, 300,
                                                        nms_cfg=dict(type="nms", iou_threshold=0.7, split_thr=split_thr))
            keep, rb = r["keep"], r["bboxes"]
        else:           # Uni order: torchvision.ops.batched_nms on network pixels (20000 coordinates: per-class branch)
            ob, os_, ol, oa, oc, oe = run_nms(L, [bx], [sc], [lb], 3, 0.7, 300, [meta], embed=emb, mode="torchvision", param=4000)
            keep = opp.torchvision_batched_nms(bx, sc, lb, 0.7, "cpu", max_keep=300)
            assert np.array_equal(keep, opp.batched_nms(bx, sc, lb, 0.7, max_keep=300))
            rb = opp.clamp_boxes(opp.rescale_boxes(bx[keep], (meta[0], meta[1]), (meta[3], meta[4])),
                                 (meta[6], meta[5]))
        assert oc[0] == keep.shape[0]
        assert np.array_equal(oa[0, :oc[0]], keep), f"pre={meta[7]}: kept set/order differs"
        assert np.array_equal(ob[0, :oc[0]], rb), f"pre={meta[7]}: output boxes differ"
        assert np.array_equal(oe[0, :oc[0]], emb[0, keep])
        assert np.all(oe[0, oc[0]:] == 0)


def _forms(opp, bx, sc, lb, thr, tv_param, split_thr, max_keep=None):
    """Oracle keeps of the three device modes on one candidate list."""
    tv_dev = {4000: "cpu", 20000: "cuda"}[tv_param]
    return dict(vanilla=opp.batched_nms(bx, sc, lb, thr, max_keep=max_keep),
                torchvision=opp.torchvision_batched_nms(bx, sc, lb, thr, tv_dev, max_keep=max_keep),
                mmcv=opp.mmcv_batched_nms(bx, sc, lb, dict(type="nms", iou_threshold=thr, split_thr=split_thr), max_keep=max_keep))


def _check_forms(L, opp, name, bx, sc, lb, k, thr=0.7, tv_param=4000, split_thr=10000, max_out=300):
    want = _forms(opp, bx, sc, lb, thr, tv_param, split_thr, max_keep=max_out)
    for mode, param in (("vanilla", 0), ("torchvision", tv_param), ("mmcv", split_thr)):
        _, os_, ol, oa, oc, _ = run_nms(L, [bx], [sc], [lb], k, thr, max_out, [IDENT], mode=mode, param=param)
        keep = want[mode]
        assert oc[0] == keep.shape[0], f"{name}/{mode}: kept {oc[0]} vs oracle {keep.shape[0]}"
        assert np.array_equal(oa[0, :oc[0]], keep), f"{name}/{mode}: kept indices differ"
        assert np.array_equal(ol[0, :oc[0]], lb[keep]) and np.array_equal(os_[0, :oc[0]], sc[keep])
    return want


def test_nms_library_forms_hand_vectors_and_goldens(L):
    """torchvision.ops.batched_nms and mmcv.ops.batched_nms as published (coordinate offsets in fp32, agnostic pass,
    candidate-count branches, double vs float threshold) against the hand-derived vectors and the oracle's records."""
    from oracle import postprocess as opp
    fx = golden("nms.npz")
    for name in ("quant", "cross", "thr03"):
        bx, sc, lb, thr = (fx[f"hand.{name}.{k}"] for k in ("boxes", "scores", "labels", "thr"))
        thr = float(thr)
        k = int(lb.max()) + 1
        for mode, param, key in (("vanilla", 0, "vanilla"), ("torchvision", 4000, "tv"), ("mmcv", 10000, "mmcv"), ("mmcv", 1, "mmcv_split")):
            _, _, _, oa, oc, _ = run_nms(L, [bx], [sc], [lb], k, thr, 300, [IDENT], mode=mode, param=param)
            assert oa[0, :oc[0]].tolist() == fx[f"hand.{name}.{key}"].tolist(), f"hand vector {name}, {mode}/{param}"
    # the hand-derived expectations themselves (tests/golden/make_golden.py nms_hand_cases): quantisation keeps a box
    # the label test suppresses; a class-1 box below -1 meets a class-0 box in the agnostic pass; double vs float 0.3
    assert fx["hand.quant.tv"].tolist() == [0, 1, 2] and fx["hand.quant.vanilla"].tolist() == [0, 2]
    assert fx["hand.cross.mmcv"].tolist() == [0] and fx["hand.cross.mmcv_split"].tolist() == [0, 1]
    assert fx["hand.thr03.tv"].tolist() == [0] and fx["hand.thr03.mmcv"].tolist() == [0, 1]
    want = _check_forms(L, opp, "rand", fx["rand.boxes"], fx["rand.scores"], fx["rand.labels"], 5, max_out=1024)
    assert np.array_equal(want["mmcv"], fx["rand.keep_mmcv"][:1024]) and np.array_equal(want["vanilla"], fx["rand.keep"][:1024])
    _check_forms(L, opp, "rand/split100+trick20000", fx["rand.boxes"], fx["rand.scores"], fx["rand.labels"], 5, tv_param=20000,
                 split_thr=100, max_out=1024)
    want = _check_forms(L, opp, "lvis", fx["lvis.boxes"], fx["lvis.scores"], fx["lvis.labels"], 1203, max_out=1024)
    assert np.array_equal(want["torchvision"], fx["lvis.keep_tv"]) and np.array_equal(want["mmcv"], fx["lvis.keep_mmcv"])
    assert not np.array_equal(want["vanilla"], want["mmcv"]), "label 1100-1202 on 1280-px boxes: the offset forms must differ from the label test"


@pytest.mark.parametrize("case", ["lvis12000", "negative", "all_negative", "far_negative", "split_boundary"])
def test_nms_library_forms_seeded(L, case):
    """Seeded candidate lists that exercise every branch of the offset forms: >= split_thr candidates with LVIS-sized
    labels, coordinates below -1 (cross-class reach 1..4 buckets), all-negative boxes (offset step <= 0: whole-list
    walk), a reach beyond the bucket radius, and candidate counts at the branch boundaries."""
    from oracle import postprocess as opp
    g = np.random.default_rng({"lvis12000": 1, "negative": 2, "all_negative": 3, "far_negative": 4, "split_boundary": 5}[case])
    f = np.float32

    def boxes(n, lo, hi, wmin=8, wmax=90):
        ctr = g.random((n, 2), dtype=np.float32) * f(hi - lo) + f(lo)
        wh = g.random((n, 2), dtype=np.float32) * f(wmax - wmin) + f(wmin)
        return np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(f)

    def scores(n):
        return np.sort(np.round(g.random(n, dtype=np.float32) * 4000) / 4000)[::-1].copy()     # with exact ties
    if case == "lvis12000":
        n, k = 12000, 1203
        bx = boxes(n, 0, 1280, 20, 200)
        bx[n // 2:] = bx[:n // 2] + (g.random((n // 2, 1), dtype=np.float32) * f(24) - f(12))   # heavy overlaps, some near 0.7
        lb = np.concatenate([g.integers(0, 1203, n // 2)] * 2).astype(np.int64)
        _check_forms(L, opp, case, bx, scores(n), lb, k)
    elif case == "negative":
        for lo, nlab in ((-40, 5), (-260, 3), (-390, 4)):           # reach 1, 2-3, 3-4 with max ~ 100
            n = 1500
            bx = boxes(n, lo, 100, 10, 70)
            # plant cross-class coincidences: box j+1 = box j shifted by -(max+1) in x and y, next label
            S = f(bx.max()) + f(1)
            amax = int(np.argmax(bx.max(1)))
            src = [int(j) for j in g.choice(n, 60, replace=False) if j != amax and (j + 7) % n != amax]
            lb = g.integers(0, nlab, n).astype(np.int64)
            for j in src:
                t = (j + 7) % n
                bx[t] = bx[j] - S
                lb[t] = lb[j] + 1
            assert f(bx.max()) + f(1) == S
            sc = scores(n)
            want = _check_forms(L, opp, f"{case}{lo}", bx, sc, lb, nlab + 1, tv_param=20000)
            assert not np.array_equal(want["mmcv"], want["vanilla"]), "planted cross-class pairs must matter"
    elif case == "all_negative":
        n = 800
        bx = boxes(n, -900, -300, 10, 120)
        _check_forms(L, opp, case, bx, scores(n), g.integers(0, 9, n).astype(np.int64), 9)
    elif case == "far_negative":
        n = 900
        bx = boxes(n, -2000, 60, 10, 120)
        _check_forms(L, opp, case, bx, scores(n), g.integers(0, 40, n).astype(np.int64), 40)
    else:
        for n in (999, 1000, 1001):                      # torchvision: 4n <= 4000 -> trick, else vanilla
            bx = boxes(n, -30, 400, 10, 150)
            lb = g.integers(0, 80, n).astype(np.int64)
            _check_forms(L, opp, f"{case}{n}", bx, scores(n), lb, 80, split_thr=1000)          # mmcv: n < 1000 agnostic


# ------------------------------------------------------------------------------------------ retrieval
@pytest.mark.parametrize("k", [80, 81, 256, 1203])
def test_retrieval_max_golden(L, k):
    from oracle import postprocess as opp
    from wedetect_amd import weights as W
    fx = golden("retrieval.npz")
    e = W.make_regions(300, seed=int(fx[f"k{k}.seed_embed"]))
    assert np.array_equal(e[:4, :8], fx[f"k{k}.embed_head"]), "region generator drifted from the golden fixture"
    t = W.make_text_bank(k)
    scale, bias = fx[f"k{k}.scale"], fx[f"k{k}.bias"]
    n_img, rows = 3, 300
    counts = np.asarray([300, 0, 137], np.int32)
    ed = dev(np.stack([e, e, e]))
    out = torch.full((n_img, k), -1.0, device="cuda")
    L.retrieval_max(ed, dev(t), dev(np.stack([scale] * 3)), dev(np.stack([bias] * 3)), dev(counts, torch.int32), out,
                    n_img, rows, k, 768)
    o = to_np(out)
    assert_close(f"retrieval max k{k} (golden from the reference's lines)", o[0], fx[f"k{k}.max_scores"], 2e-6, 1e-5)
    assert np.all(o[1] == 0.0), "image without regions must give zeros"
    assert_close("partial count", o[2], opp.retrieval_scores(e[:137], t, scale[:137], bias[:137]), 2e-6, 1e-5)


def test_nms_iou_is_exactly_the_fp32_formula_near_the_threshold(L):
    """4000 two-box images whose IoU sits within a few fp32 ulps of the 0.7 threshold: the keep decision must be
    the one the fp32 formula inter / (a_i + a_j - inter) gives with every operation rounded separately (numpy) —
    an fma-contracted union is off by up to 2 ulps and flips some of these."""
    g = np.random.default_rng(404)
    n = 4000
    boxes, expect = [], []
    f = np.float32
    thr = f(0.7)
    while len(boxes) < n:
        w, h = f(g.uniform(20, 200)), f(g.uniform(20, 200))
        x, y = f(g.uniform(0, 300)), f(g.uniform(0, 300))
        a = np.array([x, y, x + w, y + h], f)
        # shift along x so that IoU ~ 0.7: inter = (w - d) h, union = (w + d) h  ->  d = w (1 - t) / (1 + t)
        d = f(w * (1 - 0.7) / (1 + 0.7)) + f(g.uniform(-2e-4, 2e-4))
        b = np.array([a[0] + d, a[1], a[2] + d, a[3]], f)
        ai, aj = (a[2] - a[0]) * (a[3] - a[1]), (b[2] - b[0]) * (b[3] - b[1])
        iw = max(f(0), min(a[2], b[2]) - max(a[0], b[0])); ih = max(f(0), min(a[3], b[3]) - max(a[1], b[1]))
        inter = f(iw * ih)
        iou = inter / f(f(ai + aj) - inter)
        if abs(float(iou) - 0.7) < 3e-6:
            boxes.append(np.stack([a, b]))
            expect.append(1 if iou > thr else 2)
    sc = [np.array([0.9, 0.8], f)] * n
    lb = [np.zeros(2, np.int64)] * n
    _, _, _, _, oc, _ = run_nms(L, boxes, sc, lb, 1, 0.7, 4, [IDENT] * n)
    expect = np.asarray(expect)
    assert 0.2 < np.mean(expect == 1) < 0.8, "the construction must straddle the threshold"
    assert np.array_equal(oc, expect), f"{int(np.sum(oc != expect))} of {n} near-threshold decisions differ"


def test_time_next_gemm_stamps_exactly_one_launch():
    """wd_time_next_gemm: the events carry the kernel's own begin / end (no stream barriers), for one launch only;
    the reading agrees with (and never exceeds) an event pair recorded around the same launch."""
    from wedetect_amd import lib as L
    m, n, k = 65536, 256, 512
    a = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, device="cuda") * k ** -0.5
    c = torch.empty(m, n, device="cuda")
    ws = L.split_weights(w)
    kw = dict(batch=1, hin=1, win=m, cin=k, lda=k, n=n, ldc=n)
    for split in (None, ws):
        run = lambda: L.conv_gemm(a, None if split else w, None, c, w_split=split, **kw)
        for _ in range(3):
            run()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for e in ev:
            e.record()                                     # materialise the hipEvent_t handles
        torch.cuda.synchronize()
        ev[0].record()
        L.time_next_gemm(ev[1], ev[2])
        run()                                              # stamped
        ev[3].record()
        run()                                              # not stamped: the hook is one-shot
        torch.cuda.synchronize()
        inner, outer = ev[1].elapsed_time(ev[2]), ev[0].elapsed_time(ev[3])
        flops = 2.0 * m * n * k
        assert 0.0 < inner <= outer * 1.02, (inner, outer)
        assert inner > outer * 0.5, (inner, outer)         # same launch: the bracket adds microseconds, not multiples
        assert flops / (inner * 1e-3) / 1e12 < (900.0 if split else 160.0)   # below the MFMA roofs: a real duration
        L.time_next_gemm(ev[4], ev[5])
        L.check(L.LIB.wd_time_next_gemm(None, None), "clear")
        run()                                              # cleared: ev[4], ev[5] keep their old stamps
        torch.cuda.synchronize()
        assert abs(ev[4].elapsed_time(ev[5])) < 1.0


@pytest.mark.parametrize("b,h,w,c", [(2, 20, 20, 1024), (1, 40, 40, 512), (3, 17, 23, 96), (1, 8, 16, 32), (2, 33, 9, 192), (1, 5, 5, 1536),
                                     (3, 160, 160, 128), (1, 37, 53, 128), (4, 20, 36, 64), (2, 16, 16, 128),
                                     # round 5: the wide register form (c = 256 / 384 / 512: channel blocks dealt to one or two thread
                                     # groups, the LayerNorm butterfly's last block step through the LDS): BASELINE stage shapes, ragged
                                     # tiles, both sides of the four-rows-per-group LayerNorm threshold at c = 256 (rows >= 65536)
                                     (32, 40, 40, 512), (3, 40, 40, 512), (2, 13, 21, 512), (33, 80, 80, 256), (2, 80, 80, 256), (1, 9, 7, 256),
                                     (2, 40, 40, 384), (1, 11, 19, 384)])
@pytest.mark.parametrize("split", [False, True])
def test_dwconv7_ln_fused_bit_identical_to_the_pair(b, h, w, c, split):
    """wd_dwconv7_ln == wd_dwconv7 followed by wd_layernorm_rows(_split) in place, bit for bit.  c > 128: one workgroup = an
    8 x 16 pixel tile over ALL channels, rows normalised from L2 right after they were written.  c <= 128 (round 3): a
    16 x 16 tile whose pre-norm values stay in registers, the LayerNorm butterfly done as register adds over the channel
    blocks + three lane exchanges.  Ragged tiles, every LN lane-group shape (8 / 16 / 32 lanes x 1 quad, NV 1..6), row counts
    on both sides of the four-rows-per-group LayerNorm threshold (65536), fp32 and fp16 hi/lo outputs.  Round 5: c = 256 / 384 /
    512 take the wide register form (dwconv7_ln_wide_kernel)."""
    from wedetect_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(c + h)
    x = torch.randn(b * h * w, c, device="cuda", generator=g) * 2.0
    w7 = torch.randn(49, c, device="cuda", generator=g) * 0.15
    bias = torch.randn(c, device="cuda", generator=g) * 0.1
    gam = torch.rand(c, device="cuda", generator=g) + 0.5
    bet = torch.randn(c, device="cuda", generator=g) * 0.1
    ref = torch.empty_like(x)
    L.dwconv7(x, w7, bias, ref, b, h, w, c)
    L.layernorm_rows(ref, ref, gam, bet, b * h * w, c, split=split)
    for _ in range(2):
        out = torch.full_like(x, 3.0)
        L.dwconv7_ln(x, w7, bias, out, gam, bet, b, h, w, c, split=split)
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), float((out - ref).abs().max())
    with pytest.raises(L.WedetectHipError):
        L.dwconv7_ln(x[:, :24].contiguous(), w7[:, :24].contiguous(), bias[:24], out[:, :24].contiguous(), gam[:24], bet[:24], b, h, w, 24)


@pytest.mark.parametrize("b,h,w,c,n", [(2, 16, 24, 128, 256), (1, 40, 40, 256, 512), (3, 6, 10, 64, 96), (32, 20, 20, 64, 256)])
def test_downsample_as_plain_gemm_on_space_to_depth_layernorm_rows(b, h, w, c, n):
    """ConvNeXt downsample (LayerNorm2d -> Conv2d k2 s2, mm_backbone.py downsample_layers): wd_layernorm_rows_split_s2d
    writes the convolution's GEMM rows ((kh, kw, cin) columns), so the layer runs as a plain pre-split GEMM with K = 4 c.
    Same operands in the same K order: bit-identical to LayerNorm + the 2 x 2 / stride-2 conv kernel, and equal to torch."""
    from wedetect_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(b * 1000 + c)
    x = torch.randn(b * h * w, c, device="cuda", generator=g) * 1.5 + 0.2
    gam = torch.rand(c, device="cuda", generator=g) + 0.5
    bet = torch.randn(c, device="cuda", generator=g) * 0.1
    wt = torch.randn(n, c, 2, 2, device="cuda", generator=g) * (4 * c) ** -0.5
    bias = torch.randn(n, device="cuda", generator=g) * 0.1
    w_rows = wt.permute(0, 2, 3, 1).reshape(n, 4 * c).contiguous()          # (kh, kw, cin): pack._conv_rows
    ws = L.split_weights(w_rows)
    m_out = b * (h // 2) * (w // 2)
    # conv path
    t1 = torch.empty(b * h * w, c, device="cuda")
    L.layernorm_rows(x, t1, gam, bet, b * h * w, c, split=True)
    y1 = torch.empty(m_out, n, device="cuda")
    L.conv_gemm(t1, None, bias, y1, w_split=ws, batch=b, hin=h, win=w, cin=c, lda=c, kh=2, kw=2, stride=2, pad=0, n=n, ldc=n,
                split_flags=L.SPLIT_A)
    # space-to-depth path
    t2 = torch.empty(m_out, 4 * c, device="cuda")
    L.layernorm_rows_split_s2d(x, t2, gam, bet, b, h, w, c)
    y2 = torch.empty(m_out, n, device="cuda")
    L.conv_gemm(t2, None, bias, y2, w_split=ws, batch=b, hin=h // 2, win=w // 2, cin=4 * c, lda=4 * c, n=n, ldc=n,
                split_flags=L.SPLIT_A)
    torch.cuda.synchronize()
    assert torch.equal(y1.view(torch.int32), y2.view(torch.int32)), float((y1 - y2).abs().max())
    ln = F.layer_norm(x.double(), (c,), gam.double(), bet.double(), 1e-6).view(b, h, w, c).permute(0, 3, 1, 2)
    ref = F.conv2d(ln, wt.double(), bias.double(), stride=2).permute(0, 2, 3, 1).reshape(m_out, n)
    assert_close("downsample s2d", y2, ref.float().cpu().numpy(), 2e-5, 2e-5)
    with pytest.raises(L.WedetectHipError):
        L.layernorm_rows_split_s2d(x[: 3 * 5 * c // c], t2, gam, bet, 1, 3, 5, c)          # odd map
