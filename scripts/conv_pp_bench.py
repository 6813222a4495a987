"""Per-shape A/B of the neck / head convolution family at the benchmark batch: register-staged loader-split kernel on
fp32 activations (rounds 1-2) vs the LDS-DMA implicit-GEMM kernel on pre-split activations (split_gemm_conv.hip), ring
depth 3 / 4.  Prints us per launch (median of 20 after 5 warm-ups, HIP events) and checks bit-identity per line."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L

SHAPES = [  # b, h, w, cin, n, k, stride, res, out
    (32, 40, 40, 128, 128, 3, 1, True, "split"),
    (32, 40, 40, 128, 128, 3, 1, False, "split"),
    (32, 80, 80, 64, 64, 3, 1, True, "split"),
    (32, 20, 20, 256, 256, 3, 1, True, "split"),
    (32, 80, 80, 128, 128, 3, 2, False, "split"),
    (32, 40, 40, 128, 128, 3, 2, False, "split"),
    (32, 40, 40, 256, 256, 3, 2, False, "split"),
    (32, 80, 80, 128, 256, 3, 1, False, "split"),
    (32, 80, 80, 256, 256, 3, 1, False, "split"),
    (32, 80, 80, 128, 64, 3, 1, False, "split"),
    (32, 40, 40, 256, 256, 3, 1, False, "split"),
    (32, 20, 20, 512, 256, 3, 1, False, "split"),
    (32, 80, 80, 256, 768, 1, 1, False, "f32"),
    (32, 40, 40, 256, 128, 1, 1, False, "split"),
    (32, 80, 80, 128, 64, 1, 1, False, "split"),
]


def to_split(x):
    rows, k = x.shape
    out = torch.empty(rows, k, device="cuda")
    L.check(L.LIB.wd_split_weights(x.data_ptr(), rows, k, 1.0, out.data_ptr(), L.stream_ptr()), "split")
    return out


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    work = torch.empty(2 * 32 * 400 * 256 + 64, device="cuda")
    print(f"{'shape':44s} {'old us':>8s} {'TF':>6s} | {'dma auto':>8s} {'TF':>6s} {'ring 3':>8s} {'ring 4':>8s}  identical")
    for (b, h, w, ci, n, k, s, res, out) in SHAPES:
        pad = 1 if k == 3 else 0
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        m = b * ho * wo
        x = torch.randn(b * h * w, ci, device="cuda", generator=g)
        wt = torch.randn(n, k * k * ci, device="cuda", generator=g) * (k * k * ci) ** -0.5
        bias = torch.randn(n, device="cuda", generator=g)
        r = torch.randn(m, n, device="cuda", generator=g) if res else None
        ws = L.split_weights(wt)
        xs = to_split(x)
        ksp = 2 if (k == 3 and s == 1 and h * w <= 400 and 9 * ci >= 2304) else 0
        geo = dict(batch=b, hin=h, win=w, cin=ci, lda=ci, kh=k, kw=k, stride=s, pad=pad, n=n, ldc=n, act=L.ACT_SILU, res=r,
                   ldres=n if res else 0, res_alpha=0.5, w_split=ws)
        if ksp:
            geo.update(workspace=work, k_splits=ksp)
        c0, c1 = torch.empty(m, n, device="cuda"), torch.empty(m, n, device="cuda")
        fl = L.SPLIT_A | (L.SPLIT_C if out == "split" else 0)
        t_old = timeit(lambda: L.conv_gemm(x, None, bias, c0, **geo))
        row = []
        for cfg in (-1, 73, 74):
            try:
                row.append(timeit(lambda: L.conv_gemm(xs, None, bias, c1, split_flags=fl, split_cfg=cfg, **geo)))
            except Exception as e:
                row.append(float("nan"))
        L.conv_gemm(xs, None, bias, c1, split_flags=fl, split_cfg=-1, **geo)
        same = torch.equal(c1.view(torch.int32), (to_split(c0) if out == "split" else c0).view(torch.int32))
        fl_ = 2.0 * m * n * k * k * ci
        print(f"{b}x{h}x{w} c{ci}->{n} {k}x{k} s{s} res={int(res)} ks={ksp} {out:5s}".ljust(44) +
              f" {t_old:8.1f} {fl_ / t_old / 1e6:6.1f} | {row[0]:8.1f} {fl_ / row[0] / 1e6:6.1f} {row[1]:8.1f} {row[2]:8.1f}  {same}")


if __name__ == "__main__":
    main()
