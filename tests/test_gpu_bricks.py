"""Row f4: the text-guided attention bricks (wedetect/models/layers/yolo_bricks.py:161-243, 572-648) on the
device vs (a) the committed outputs of the reference classes themselves (tests/golden/bricks.npz) and (b) the
oracle restatement on fresh seeded inputs, including shapes the fixture does not hold (1203 guide rows, 128-wide
heads, 80 x 80 maps).  Tolerance 1e-3 (north_star), typical error 1e-6."""
import json

import numpy as np
import pytest
import torch

from tests.util import assert_close, golden

pytestmark = pytest.mark.gpu

PRECISIONS = ("fp32", "fp16x3")


def _params(fx, prefix):
    return {k[len(prefix) + 3:]: fx[k] for k in fx.files if k.startswith(prefix + ".p.")}


@pytest.mark.parametrize("precision", PRECISIONS)
def test_max_sigmoid_attn_matches_reference_fixture(precision):
    from wedetect_amd.bricks import MaxSigmoidAttnBlock
    fx = golden("bricks.npz")
    for j in range(3):
        c = json.loads(str(fx[f"msa{j}.cfg"]))
        kw = {k: c[k] for k in ("in_channels", "out_channels", "guide_channels", "embed_channels", "num_heads", "with_scale")}
        m = MaxSigmoidAttnBlock(**kw, precision=precision).load_state_dict(_params(fx, f"msa{j}"))
        out = m(torch.from_numpy(fx[f"msa{j}.x"]).cuda(), torch.from_numpy(fx[f"msa{j}.guide"]).cuda())
        assert out.shape == fx[f"msa{j}.out"].shape
        err = assert_close(f"msa{j}[{precision}]", out, fx[f"msa{j}.out"], 1e-3)
        print(f"msa{j} {precision}: max|d| {err:.2e}")


@pytest.mark.parametrize("precision", PRECISIONS)
def test_image_pooling_attention_matches_reference_fixture(precision):
    from wedetect_amd.bricks import ImagePoolingAttentionModule
    fx = golden("bricks.npz")
    for j in range(2):
        c = json.loads(str(fx[f"ipa{j}.cfg"]))
        kw = {k: c[k] for k in ("image_channels", "text_channels", "embed_channels", "num_heads", "with_scale")}
        m = ImagePoolingAttentionModule(**kw, precision=precision).load_state_dict(_params(fx, f"ipa{j}"))
        feats = [torch.from_numpy(fx[f"ipa{j}.feat{l}"]).cuda() for l in range(len(c["image_channels"]))]
        out = m(torch.from_numpy(fx[f"ipa{j}.text"]).cuda(), feats)
        err = assert_close(f"ipa{j}[{precision}]", out, fx[f"ipa{j}.out"], 1e-3)
        print(f"ipa{j} {precision}: max|d| {err:.2e}")


def _rand_params(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, s in shapes.items():
        if k.endswith("running_var"):
            p[k] = torch.rand(s, generator=g) + 0.5
        elif k.endswith(("bn.weight", ".0.weight")):
            p[k] = torch.rand(s, generator=g) + 0.5
        elif k == "scale":
            p[k] = torch.rand(s, generator=g) + 0.25
        elif len(s) > 1:
            p[k] = torch.randn(s, generator=g) / float(np.prod(s[1:])) ** 0.5
        else:
            p[k] = torch.randn(s, generator=g) * 0.2
    return p


@pytest.mark.parametrize("cin,cout,embed,heads,gch,n,b,h,w,with_scale", [
    (128, 128, 128, 4, 768, 80, 2, 80, 80, False),      # YOLO-World-sized P3 block: 32-wide heads, embed_conv absent
    (64, 256, 256, 2, 512, 1203, 1, 20, 20, True),      # 128-wide heads, LVIS-sized guide: several LDS passes
    (96, 64, 64, 8, 32, 3, 3, 17, 5, True),             # 8-wide heads, ragged map
])
def test_max_sigmoid_attn_vs_oracle(cin, cout, embed, heads, gch, n, b, h, w, with_scale):
    from oracle import bricks as obr
    from wedetect_amd.bricks import MaxSigmoidAttnBlock
    shapes = {"guide_fc.weight": (embed, gch), "guide_fc.bias": (embed,), "bias": (heads,),
              "project_conv.conv.weight": (cout, cin, 3, 3)}
    for nm, ch in (("project_conv", cout),) + ((("embed_conv", embed),) if embed != cin else ()):
        shapes.update({f"{nm}.bn.weight": (ch,), f"{nm}.bn.bias": (ch,), f"{nm}.bn.running_mean": (ch,), f"{nm}.bn.running_var": (ch,)})
    if embed != cin:
        shapes["embed_conv.conv.weight"] = (embed, cin, 1, 1)
    if with_scale:
        shapes["scale"] = (1, heads, 1, 1)
    p = _rand_params(shapes, 5 + cin)
    g = torch.Generator().manual_seed(9)
    x, guide = torch.randn(b, cin, h, w, generator=g), torch.randn(b, n, gch, generator=g)
    ref = obr.max_sigmoid_attn(x, guide, p, heads)
    for precision in PRECISIONS:
        m = MaxSigmoidAttnBlock(cin, cout, gch, embed, num_heads=heads, with_scale=with_scale, precision=precision).load_state_dict(p)
        out = m(x.cuda(), guide.cuda())
        err = assert_close(f"msa {cin}->{cout} heads {heads} n {n} [{precision}]", out, ref, 1e-3)
        print(f"msa {cin}->{cout} heads {heads} n {n} {precision}: max|d| {err:.2e}")


@pytest.mark.parametrize("chans,ct,embed,heads,n,b,sizes,with_scale", [
    ([128, 256, 512], 768, 256, 8, 80, 2, [(80, 80), (40, 40), (20, 20)], True),     # the reference's default geometry
    ([32, 32, 64], 64, 128, 2, 130, 1, [(10, 7), (5, 4), (3, 3)], False),            # 64-wide heads, > 64 (and > 128) text rows
])
def test_image_pooling_attention_vs_oracle(chans, ct, embed, heads, n, b, sizes, with_scale):
    from oracle import bricks as obr
    from wedetect_amd.bricks import ImagePoolingAttentionModule
    shapes = {"proj.weight": (ct, embed), "proj.bias": (ct,)}
    for l, ch in enumerate(chans):
        shapes.update({f"projections.{l}.conv.weight": (embed, ch, 1, 1), f"projections.{l}.conv.bias": (embed,)})
    for nm, d in (("query", ct), ("key", embed), ("value", embed)):
        shapes.update({f"{nm}.0.weight": (d,), f"{nm}.0.bias": (d,), f"{nm}.1.weight": (embed, d), f"{nm}.1.bias": (embed,)})
    if with_scale:
        shapes["scale"] = (1,)
    p = _rand_params(shapes, 31 + ct)
    g = torch.Generator().manual_seed(13)
    text = torch.randn(b, n, ct, generator=g)
    feats = [torch.randn(b, ch, hh, ww, generator=g) for ch, (hh, ww) in zip(chans, sizes)]
    ref = obr.image_pooling_attention(text, feats, p, heads)
    for precision in PRECISIONS:
        m = ImagePoolingAttentionModule(chans, ct, embed, with_scale=with_scale, num_heads=heads, precision=precision).load_state_dict(p)
        out = m(text.cuda(), [f.cuda() for f in feats])
        err = assert_close(f"ipa E {embed} heads {heads} n {n} [{precision}]", out, ref, 1e-3)
        print(f"ipa E {embed} heads {heads} n {n} {precision}: max|d| {err:.2e}")


def test_adaptive_maxpool_and_cross_attention_kernels_vs_torch():
    """The two stand-alone kernels against the torch ops the reference calls (exact for the pooling)."""
    import torch.nn.functional as F
    from wedetect_amd import lib as L
    g = torch.Generator().manual_seed(3)
    for (b, h, w, c, p) in [(2, 13, 11, 64, 3), (1, 3, 3, 8, 3), (3, 2, 5, 1024, 2), (1, 40, 40, 256, 3), (2, 2, 2, 16, 3)]:
        x = torch.randn(b, c, h, w, generator=g)
        rows = x.permute(0, 2, 3, 1).contiguous().view(b * h * w, c).cuda()
        out = torch.empty(b * p * p, c, device="cuda")
        L.adaptive_maxpool_nhwc(rows, out, p * p * c, b, h, w, c, p)
        ref = F.adaptive_max_pool2d(x, (p, p)).permute(0, 2, 3, 1).reshape(b * p * p, c)
        assert torch.equal(out.cpu(), ref), (b, h, w, c, p)
    for (b, nq, nk, heads, dh) in [(2, 5, 27, 4, 16), (1, 200, 64, 2, 64), (3, 1, 1, 8, 8), (2, 80, 27, 8, 32)]:
        q, k, v = (torch.randn(b, n, heads, dh, generator=g) for n in (nq, nk, nk))
        a = F.softmax(torch.einsum("bnmc,bkmc->bmnk", q, k) / dh ** 0.5, dim=-1)
        ref = torch.einsum("bmnk,bkmc->bnmc", a, v).reshape(b * nq, heads * dh)
        out = torch.empty(b * nq, heads * dh, device="cuda")
        L.cross_attention_small(q.view(b * nq, -1).cuda(), k.view(b * nk, -1).cuda(), v.view(b * nk, -1).cuda(), out, b, nq, nk, heads, dh)
        assert_close(f"cross attention {(b, nq, nk, heads, dh)}", out, ref, 2e-6, 1e-5)


def test_bricks_reject_bad_arguments():
    from wedetect_amd import lib as L
    from wedetect_amd.bricks import MaxSigmoidAttnBlock
    e = torch.zeros(64, 24, device="cuda")
    with pytest.raises(L.WedetectHipError):                      # 12-wide heads: unsupported head width
        L.max_sigmoid_attn(e, torch.zeros(1, 2, 24, device="cuda"), torch.zeros(2, device="cuda"), None, e.clone(), 1, 64, 2, 2, 12, 12)
    with pytest.raises(L.WedetectHipError):                      # more than 64 keys
        L.cross_attention_small(torch.zeros(4, 16, device="cuda"), torch.zeros(65, 16, device="cuda"), torch.zeros(65, 16, device="cuda"),
                                torch.zeros(4, 16, device="cuda"), 1, 4, 65, 1, 16)
    with pytest.raises(NotImplementedError):
        MaxSigmoidAttnBlock(8, 8, 8, 8, use_depthwise=True)
    with pytest.raises(RuntimeError):
        MaxSigmoidAttnBlock(8, 8, 8, 8)(torch.zeros(1, 8, 4, 4, device="cuda"), torch.zeros(1, 1, 8, device="cuda"))
