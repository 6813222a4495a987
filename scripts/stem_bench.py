"""The stem (uint8 image -> / 255 -> 4 x 4 s4 conv -> LayerNorm): three launches against wd_stem_fused, HIP-event timed.
    PYTHONPATH=. python scripts/stem_bench.py [batch] [size] [c0]"""
import sys
import torch
from wedetect_amd import lib as L

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 640
c0 = int(sys.argv[3]) if len(sys.argv) > 3 else 128
g = torch.Generator(device="cuda").manual_seed(0)
img = torch.randint(0, 256, (b, hw, hw, 3), dtype=torch.uint8, device="cuda", generator=g)
r = lambda *s, k=1.0: torch.randn(*s, device="cuda", generator=g) * k
wt, bias, gm, bt = r(c0, 48, k=0.3), r(c0, k=0.1), r(c0, k=0.2) + 1.0, r(c0, k=0.1)
m = b * (hw // 4) ** 2
patches, x3, x1 = torch.empty(m, 48, device="cuda"), torch.empty(m, c0, device="cuda"), torch.empty(m, c0, device="cuda")


def three():
    L.stem_patchify(img, patches)
    L.conv_gemm(patches, wt, bias, x3, batch=b, hin=hw // 4, win=hw // 4, cin=48, lda=48, n=c0, ldc=c0)
    L.layernorm_rows(x3, x3, gm, bt, m, c0)


def fused():
    L.stem_fused(img, wt, bias, gm, bt, x1)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


three(); fused(); torch.cuda.synchronize()
print("bit-identical:", torch.equal(x3.view(torch.int32), x1.view(torch.int32)))
hbm = img.numel() + m * c0 * 4
for _ in range(2):
    t3, t1 = timeit(three), timeit(fused)
    print(f"three launches {t3:7.1f} us   fused {t1:7.1f} us   ({hbm / t1 / 1e6:.2f} TB/s of image-in + rows-out)")
