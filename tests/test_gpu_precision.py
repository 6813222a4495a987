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


def _rescaled_checkpoint(arch, stream_scale, hidden_scale=1.0):
    """A checkpoint whose residual streams c1..c4 live at ``stream_scale`` (per channel: an array works too) and whose
    MLP hidden activations live at ``hidden_scale`` — with the layers that read them compensated, so that everything
    downstream is O(1) again, as a trained network with such internal scales would be.  Same function of the input in
    exact arithmetic for every choice of scales up to the GELU's nonlinearity (hidden_scale changes the function, the
    fp32 tower is the reference either way)."""
    from wedetect_amd import weights as W
    from wedetect_amd.arch import get_arch
    a = get_arch(arch)
    sd = W.make_state_dict(arch)
    bb = "backbone.image_model.model."

    def chan(scale, c):
        s = np.asarray(scale, np.float32)
        return np.resize(s, c).astype(np.float32) if s.ndim else np.full(c, s, np.float32)
    for i, c in enumerate(a.dims):
        s = chan(stream_scale, c)
        if i == 0:
            sd[bb + "downsample_layers.0.1.weight"] = sd[bb + "downsample_layers.0.1.weight"] * s      # stem LayerNorm affine
            sd[bb + "downsample_layers.0.1.bias"] = sd[bb + "downsample_layers.0.1.bias"] * s
        else:
            sd[bb + f"downsample_layers.{i}.1.weight"] = sd[bb + f"downsample_layers.{i}.1.weight"] * s[:, None, None, None]
            sd[bb + f"downsample_layers.{i}.1.bias"] = sd[bb + f"downsample_layers.{i}.1.bias"] * s
        for j in range(a.depths[i]):
            q = bb + f"stages.{i}.{j}."
            sd[q + "gamma"] = sd[q + "gamma"] * s
            sd[q + "dwconv.weight"] = sd[q + "dwconv.weight"] / s[:, None, None, None]       # LayerNorm input back to O(1): keeps eps's role
            sd[q + "pwconv1.weight"] = sd[q + "pwconv1.weight"] * np.float32(hidden_scale)
            sd[q + "pwconv1.bias"] = sd[q + "pwconv1.bias"] * np.float32(hidden_scale)
            sd[q + "pwconv2.weight"] = sd[q + "pwconv2.weight"] / np.float32(hidden_scale)
    for name, i in (("reduce_layer0", 3), ("Bifusion0.cv1", 2), ("Bifusion0.cv2", 1), ("Bifusion1.cv1", 1), ("Bifusion1.cv2", 0)):
        k = f"neck.{name}.block.conv.weight"
        sd[k] = sd[k] / chan(stream_scale, a.dims[i])[None, :, None, None]
    return sd


@pytest.mark.parametrize("case", ["streams_1e-4", "hidden_1e-3", "streams_3e4", "mixed_channels"])
def test_fp16x3_low_and_mixed_range_checkpoints(case):
    """fp16 (hi, lo) pairs carry an fp32 value to 2^-22 RELATIVE only while lo is a normal fp16 number
    (|x| >~ 2^-3); below that the pair has an ABSOLUTE error floor of ~3e-8, i.e. a tensor living at 1e-4 would be carried
    to 3e-4 relative.  ImageTower.calibrate() (one fp32 pass on the first batch) gives every split tensor outside
    [2^-3, 2^13] a power-of-two scale applied by its producer and divided out by its consumer.  Checked here against the
    fp32-MFMA tower on checkpoints whose residual streams / hidden activations are tiny, huge, or mixed per channel:
      * uniformly small or large tensors: calibrated fp16x3 is as close to fp32 as on ordinary checkpoints (<= 3e-5 of the
        embeddings' rms), where the uncalibrated path is 10-100x worse or overflows;
      * per-channel mixed scales inside ONE tensor (1e-4 next to 1e+1, five decades): with the maximum placed at 2^10 the
        small channels' low halves stay normal down to 2^-13 of the maximum — the same bound holds."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    arch, b, hw = "base", 2, 128
    if case == "streams_1e-4":
        sd = _rescaled_checkpoint(arch, 1e-4)
    elif case == "hidden_1e-3":
        sd = _rescaled_checkpoint(arch, 1.0, hidden_scale=1e-3)
    elif case == "streams_3e4":
        sd = _rescaled_checkpoint(arch, 3e4)                  # beyond the fp16 range without a scale: the guard would trip
    else:
        sd = _rescaled_checkpoint(arch, np.asarray([1e-4, 1e1], np.float32))
    packed = pack(sd, arch)
    x = torch.from_numpy(W.make_images(b, hw, hw, seed=7)).cuda()
    ref = ImageTower(arch, packed, b, hw, hw, precision="fp32")
    e_ref = ref.features(x)[0].clone()
    rms = float(e_ref.pow(2).mean().sqrt())
    raw = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    e_raw = raw.features(x)[0].clone()
    tripped = bool(raw.range_flags.any())
    err_raw = float((e_raw - e_ref).abs().max()) / rms if not tripped else float("inf")
    cal = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    amax = cal.calibrate(x)
    e_cal = cal.features(x)[0].clone()
    assert not bool(cal.range_flags.any()), f"{case}: range guard tripped after calibration"
    err_cal = float((e_cal - e_ref).abs().max()) / rms
    print(f"[range] {case}: embeddings rms {rms:.3g}; max err / rms: uncalibrated {err_raw:.3g}"
          f"{' (range guard tripped)' if tripped else ''}, calibrated {err_cal:.3g}; {len(cal.sscale)} of {len(amax)} tensors rescaled")
    assert len(cal.sscale) > 0, "the calibration must have found tensors outside the window"
    assert err_cal <= 3e-5, f"{case}: calibrated fp16x3 is {err_cal:.3g} of rms away from fp32"
    assert tripped or err_raw > 3.0 * err_cal, f"{case}: the case must actually stress the uncalibrated path ({err_raw:.3g} vs {err_cal:.3g})"


def test_calibration_on_an_ordinary_checkpoint_changes_nothing_above_the_fp32_noise():
    """Power-of-two scales are exact: on the synthetic checkpoints (every tensor O(1)) the calibrated tower differs from
    the uncalibrated one only through elements whose low halves were subnormal before — by less than the distance of
    either from the fp32 tower."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    arch, b, hw = "base", 1, 128
    packed = pack(W.make_state_dict(arch), arch)
    x = torch.from_numpy(W.make_images(b, hw, hw)).cuda()
    e32 = ImageTower(arch, packed, b, hw, hw, precision="fp32").features(x)[0].clone()
    t0 = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    e0 = t0.features(x)[0].clone()
    t1 = ImageTower(arch, packed, b, hw, hw, precision="fp16x3")
    amax = t1.calibrate(x)
    assert len(amax) > 100 and len(t1.sscale) > 50 and all(v == 2.0 ** round(np.log2(v)) for v in t1.sscale.values())
    assert all(2.0 ** 9 <= amax[k] * t1.sscale.get(k, 1.0) < 2.0 ** 10 for k in amax if amax[k] > 0)
    e1 = t1.features(x)[0].clone()
    assert not bool(t1.range_flags.any())
    # the diagnostic fp32 views of the (scaled, split) P3..P5 buffers divide the scale out again (they used to
    # return the stored value, off by the calibrated power of two whenever it is not 1)
    assert any(t1.sscale.get(k, 1.0) != 1.0 for k in ("p3", "p4", "p5")), "the case must give the pyramid buffers a non-unit scale"
    for i, (p0, p1) in enumerate(zip(t0.pyramid(), t1.pyramid())):
        rms = float(p0.double().pow(2).mean().sqrt())
        assert float((p0 - p1).abs().max()) <= 2e-5 * max(1.0, rms), f"P{i + 3}: calibrated view differs from the uncalibrated one"
    d01, d0, d1 = float((e0 - e1).abs().max()), float((e0 - e32).abs().max()), float((e1 - e32).abs().max())
    print(f"[range] ordinary checkpoint: |calibrated - uncalibrated| {d01:.3g}; vs fp32: uncalibrated {d0:.3g}, calibrated {d1:.3g}")
    assert d01 <= 2e-5 and d1 <= max(2e-5, 1.5 * d0)


def _heavy_tail_checkpoint(arch, n_out=4, mag=1e3, gamma=0.1):
    """A "trained-like" checkpoint (VERDICT r4 weak #3 / next #5d): a few channels of every residual stream carry values
    ``mag`` x the median — the outlier channels real ConvNeXt checkpoints develop — fed by the stem LayerNorm / the
    downsample convs and kept alive through the stage by the blocks' own updates; layer scale ``gamma`` ~ 0.1 (initialised
    at 1e-6 and grown by training, mm_backbone.py:106-108).  NOTHING inside the backbone is compensated: the block
    LayerNorms see the outliers (the ordinary channels of a normalised row shrink to ~1 / (0.18 mag)), pwconv1 sums over
    them.  Only the five neck layers that read c1..c4 divide the outlier columns back, as weights trained on such streams
    would."""
    from wedetect_amd import weights as W
    from wedetect_amd.arch import get_arch
    a = get_arch(arch)
    sd = W.make_state_dict(arch)
    bb = "backbone.image_model.model."
    scales = []
    for i, c in enumerate(a.dims):
        s = np.ones(c, np.float32)
        s[[c // 7, c // 3, c // 2 + 1, c - 5][:n_out]] = mag
        scales.append(s)
        if i == 0:
            sd[bb + "downsample_layers.0.1.weight"] = sd[bb + "downsample_layers.0.1.weight"] * s
            sd[bb + "downsample_layers.0.1.bias"] = sd[bb + "downsample_layers.0.1.bias"] * s
        else:
            sd[bb + f"downsample_layers.{i}.1.weight"] = sd[bb + f"downsample_layers.{i}.1.weight"] * s[:, None, None, None]
            sd[bb + f"downsample_layers.{i}.1.bias"] = sd[bb + f"downsample_layers.{i}.1.bias"] * s
        for j in range(a.depths[i]):
            q = bb + f"stages.{i}.{j}."
            sd[q + "gamma"] = (np.sign(sd[q + "gamma"]) * np.float32(gamma) * s).astype(np.float32)
    for name, i in (("reduce_layer0", 3), ("Bifusion0.cv1", 2), ("Bifusion0.cv2", 1), ("Bifusion1.cv1", 1), ("Bifusion1.cv2", 0)):
        k = f"neck.{name}.block.conv.weight"
        sd[k] = sd[k] / scales[i][None, :, None, None]
    return sd


def test_fp16x3_on_a_heavy_tailed_checkpoint_at_base_640():
    """fp16x3 has only met the O(1) synthetic generator; real residual streams carry outlier channels 10^2 - 10^3 x the
    median, and the split scales are per TENSOR.  Base @ 640, four channels per stream at 10^3 x: the calibrated fp16x3 tower
    must stay on the fp16x3 kernels (no range-guard trip, neither flag) and within the north-star tolerance of the fp32
    tower — embeddings <= 1e-3 of their rms, scores <= 1e-3 absolute — through the detectors' own guard path
    (checked_counts).  The per-tensor scale holds because a tensor's maximum is placed at 2^10 and the halves stay normal
    down to 2^-13 of it (8 000 x): three decades of outlier are inside the window; the measured error is printed."""
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    arch, b, hw, k = "base", 1, 640, 80
    packed = pack(_heavy_tail_checkpoint(arch), arch)
    x = torch.from_numpy(W.make_images(b, hw, hw, seed=11)).cuda()
    bank = torch.from_numpy(W.make_text_bank(k)).cuda()
    ref = ImageTower(arch, packed, b, hw, hw, max_classes=k, precision="fp32")
    meta = ref.identity_meta()
    meta[:, 7] = 1.0
    kw = dict(normalize_text=True, score_thr=0.001, with_embed=True)
    ref.detect(x, bank, meta, **kw)
    e_ref, s_ref = ref.embed.clone(), ref.scores.view(-1)[: b * ref.ntot * k].clone()
    streams = [float(t.abs().max()) / float(t.abs().median()) for t in ref.x]
    assert min(streams) > 300.0, f"the checkpoint must really be heavy-tailed: max / median of c1..c4 = {streams}"
    t = ImageTower(arch, packed, b, hw, hw, max_classes=k, precision="fp16x3")
    amax = t.calibrate(x)
    res = t.detect(x, bank, meta, **kw)
    counts = t.checked_counts(res, lambda: t.detect(x, bank, meta, **kw))
    assert t.precision == "fp16x3" and not t.overflowed and not t.neck_pin and not bool(t.range_flags.any()), \
        "the range guard tripped on the heavy-tailed checkpoint: per-tensor scales are not enough"
    rms = float(e_ref.pow(2).mean().sqrt())
    err_e = float((t.embed - e_ref).abs().max()) / rms
    err_s = float((t.scores.view(-1)[: s_ref.numel()] - s_ref).abs().max())
    print(f"[range] heavy-tailed Base@640: stream max / median {['%.0f' % v for v in streams]}; embeddings rms {rms:.3g}, max err / rms "
          f"{err_e:.3g}; scores max |d| {err_s:.3g}; {len(t.sscale)} of {len(amax)} tensors rescaled; kept {counts}")
    assert err_e <= 1e-3 and err_s <= 1e-3


def test_calibrate_keeps_the_layernorm_kernel_for_blocks_with_a_large_mean_over_std():
    """a block whose pre-norm rows have |mean| / std above engine.FOLD_MAX_MEAN_OVER_STD is NOT folded (its LayerNorm
    kernel runs, `s{i}.{j}.fold_off` travels with the split scales); the other blocks keep the fold; the tower stays within 5e-5 of
    the tower that folds nothing."""
    import numpy as np
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    arch, b, hw = "base", 1, 320
    sd = W.make_state_dict(arch, num_prompts=16)
    key = "backbone.image_model.model.stages.2.1.dwconv.bias"
    assert key in sd, [k for k in sd if "dwconv.bias" in k][:3]
    sd[key] = (sd[key] + np.float32(500.0)).astype(np.float32)      # every channel's depthwise output shifted: mean >> std over the channels
    imgs = torch.from_numpy(W.make_images(b, hw, hw)).cuda()
    t_fold, t_plain = ImageTower(arch, pack(sd, arch), b, hw, hw, precision="fp16x3"), ImageTower(arch, pack(sd, arch), b, hw, hw, precision="fp16x3")
    t_plain.ln_fold = False
    t_fold.calibrate(imgs)
    t_plain.calibrate(imgs)
    off = sorted(k for k in t_fold.sscale if k.endswith("fold_off"))
    assert off == ["s2.1.fold_off"], off
    e0, _ = t_plain.features(imgs)
    e1, _ = t_fold.features(imgs)
    torch.cuda.synchronize()
    assert not bool(t_fold.range_flags.any())
    assert float((e0 - e1).abs().max()) <= 5e-5 * max(1.0, float(e0.abs().max()))
    # a second tower of the checkpoint adopts the decision with the scales; a re-calibration never re-enables the fold
    t2 = ImageTower(arch, t_fold.P, b, hw, hw, precision="fp16x3")
    t2.adopt_scales(t_fold.sscale)
    assert t2.sscale.get("s2.1.fold_off") == 2.0
    t_fold.calibrate(imgs, merge=True)
    assert t_fold.sscale.get("s2.1.fold_off") == 2.0


def test_a_fallen_back_tower_returns_to_fp16x3_after_clean_batches(monkeypatch):
    """Round 6: trip -> fp32 fallback -> FALLBACK_RETRY clean batches -> re-calibration on the current batch ->
    fp16x3 again -> stays.  A second trip doubles the waiting time.  ``fp16x3_trips`` / ``fp16x3_retries`` count."""
    import warnings
    from wedetect_amd import weights as W
    from wedetect_amd.engine import ImageTower
    from wedetect_amd.pack import pack
    monkeypatch.setattr(ImageTower, "FALLBACK_RETRY", 3)
    arch, b, hw = "nano", 2, 128
    t = ImageTower(arch, pack(W.make_state_dict(arch, num_prompts=32), arch), b, hw, hw, precision="fp16x3")
    ok = torch.from_numpy(W.make_images(b, hw, hw, seed=5)).cuda()
    meta = t.identity_meta()
    kw = dict(normalize_text=False, score_thr=0.0, with_embed=True)

    def step(x):
        run = lambda: t.detect(x, t.P["prompts"], meta, **kw)
        return t.checked_counts(run(), run, lambda: t.calibrate(x, merge=True))
    t.calibrate(ok)
    step(ok)
    assert t.precision == "fp16x3" and t.fp16x3_trips == 0
    # force a trip the re-calibration cannot cure: raise the sticky flag by hand, with scales that do not change
    def tripping_step(x):
        run = lambda: t.detect(x, t.P["prompts"], meta, **kw)
        res = run()
        t.range_flags[0] = 1
        with warnings.catch_warnings(record=True) as wrec:
            warnings.simplefilter("always")
            counts = t.checked_counts(res, run, lambda: t.calibrate(x, merge=True))
        assert any("fp32 MFMA" in str(w.message) for w in wrec)
        return counts
    tripping_step(ok)
    assert t.precision == "fp32" and t.overflowed and t.fp16x3_trips == 1 and t._retry_after == 3
    ref = t.embed.clone()
    step(ok); step(ok)
    assert t.precision == "fp32" and t.fp16x3_retries == 0
    step(ok)                                                   # the third clean batch: back to fp16x3 for the next step
    assert t.precision == "fp16x3" and not t.overflowed and t.fp16x3_retries == 1
    step(ok)
    assert t.precision == "fp16x3" and t.fp16x3_trips == 1 and not bool(t.range_flags.any())
    assert float((t.embed - ref).abs().max()) < 1e-4           # fp16x3 again, within rounding noise of the fp32 step
    tripping_step(ok)                                          # a relapse: the wait doubles
    assert t.precision == "fp32" and t.fp16x3_trips == 2 and t._retry_after == 6
