"""3 x 3 / stride 1 convs of the neck / head at the benchmark batch: the tap-per-stage LDS-DMA kernel (cfg 70, split_gemm_conv.hip)
against the row-sharing kernel of round 4 (cfg 75, split_gemm_conv3.hip: one stage per (filter row, channel chunk) serves three
taps).  us per launch (median of 20 after 5 warm-ups, HIP events), algorithmic TFLOP/s, max |difference| / rms of the outputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L
from conv_pp_bench import to_split, timeit

SHAPES = [  # b, h, w, cin, n, res, out
    (32, 40, 40, 128, 128, True, "split"), (32, 40, 40, 128, 128, False, "split"), (32, 80, 80, 64, 64, True, "split"),
    (32, 80, 80, 128, 256, False, "split"), (32, 80, 80, 256, 256, False, "split"), (32, 80, 80, 128, 64, False, "split"),
    (32, 40, 40, 256, 256, False, "split"), (32, 40, 40, 256, 64, False, "split"), (32, 20, 20, 256, 256, True, "split"),
    (32, 20, 20, 512, 256, False, "split"), (32, 40, 40, 16, 128, False, "split"), (32, 40, 40, 64, 128, False, "split"),
]
g = torch.Generator(device="cuda").manual_seed(1)
work = torch.empty(2 * 32 * 400 * 256 + 64, device="cuda")
print(f"{'shape':40s} {'cfg70 us':>9s} {'TF':>6s} | {'cfg78 us':>9s} {'TF':>6s}   max|d|/rms")
for (b, h, w, ci, n, res, out) in SHAPES:
    m = b * h * w
    x = torch.randn(m, ci, device="cuda", generator=g)
    wt = torch.randn(n, 9 * ci, device="cuda", generator=g) * (9 * ci) ** -0.5
    bias = torch.randn(n, device="cuda", generator=g)
    r = torch.randn(m, n, device="cuda", generator=g) if res else None
    ws, xs = L.split_weights(wt), to_split(x)
    geo = dict(batch=b, hin=h, win=w, cin=ci, lda=ci, kh=3, kw=3, stride=1, pad=1, n=n, ldc=n, act=L.ACT_SILU, res=r,
               ldres=n if res else 0, res_alpha=0.5, w_split=ws, split_flags=L.SPLIT_A)
    if h * w <= 400 and 9 * ci >= 2304:                 # the engine's fixed two-way split-K on the small maps
        geo.update(workspace=work, k_splits=2)
    c0, c1 = torch.empty(m, n, device="cuda"), torch.empty(m, n, device="cuda")
    t0 = timeit(lambda: L.conv_gemm(xs, None, bias, c0, split_cfg=70, **geo))
    t1 = timeit(lambda: L.conv_gemm(xs, None, bias, c1, split_cfg=78, **geo))
    c3 = torch.empty(m, n, device="cuda")
    t3 = timeit(lambda: L.conv_gemm(xs, None, bias, c3, split_cfg=77, **geo))
    c4 = torch.empty(m, n, device="cuda")
    t4 = timeit(lambda: L.conv_gemm(xs, None, bias, c4, split_cfg=79, **geo))
    t2 = timeit(lambda: L.conv_gemm(xs, None, bias, c1, split_cfg=76, **geo)) if n % 128 else float("nan")
    torch.cuda.synchronize()
    d = float((c0 - c1).abs().max()) / float(c0.double().pow(2).mean().sqrt())
    fl = 2.0 * m * n * 9 * ci
    print(f"{b}x{h}x{w} c{ci}->{n} res={int(res)}{' ks2' if 'k_splits' in geo else ''}".ljust(40) + f" {t0:9.1f} {fl / t0 / 1e6:6.1f} | {t1:9.1f} {fl / t1 / 1e6:6.1f}   {d:.2e}   cfg76 (narrow, ring 2, two workgroups / CU) {t2:7.1f} | cfg77 (staggered) {t3:7.1f} {fl / t3 / 1e6:6.1f} same bits {bool(torch.equal(c1, c3))} | cfg79 (12 waves: 4 DMA + 8 free-running MFMA waves) {t4:7.1f} {fl / t4 / 1e6:6.1f} same bits {bool(torch.equal(c1, c4))}", flush=True)
