"""Per-LAYER times of the neck / head GEMMs (and, with ALL=1, every dense layer of the tower) at a benchmark batch, serial chain
($WEDETECT_DAG=0 is set here): every ImageTower._gemm call is bracketed by HIP events on the launch stream, median over ITERS
steps.  Prints layer name, geometry, kernel family, us, algorithmic TFLOP/s and the fraction of the fp16x3 roof (838.9 TF),
the layer's algorithmic HBM bytes (input once, weights, every output format it writes, the residual it reads) with the GB/s they
imply, and — round 6 — WHICH roof is the tighter one for the layer (MFMA: flops / 838.9 TF; HBM: bytes / 8 TB/s) with the fraction
of that roof; then totals per family.  ARCH / BATCH / SIZE / ITERS / ALL from the environment."""
import os, sys, statistics, json
os.environ.setdefault("WEDETECT_DAG", "0")
os.environ.setdefault("WEDETECT_BB_CHAINS", "1")       # one backbone chain: a bracketed launch must have the chip to itself
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wedetect_amd import lib as L, weights as W
from wedetect_amd.engine import ImageTower
from wedetect_amd.pack import pack

arch = os.environ.get("ARCH", "base")
B = int(os.environ.get("BATCH", "32"))
S = int(os.environ.get("SIZE", "640"))
iters = int(os.environ.get("ITERS", "7"))
show_all = os.environ.get("ALL", "0") == "1"
ROOF = 2516.6 / 3
HBM_PEAK = 8.0e12

tower = ImageTower(arch, pack(W.make_state_dict(arch, num_prompts=80), arch), B, S, S)
imgs = torch.from_numpy(W.make_images(B, S, S)).cuda()
tower.calibrate(imgs)
for _ in range(2):
    tower.features(imgs)
torch.cuda.synchronize()

records = {}          # call index -> dict
order = []
orig = ImageTower._gemm
state = {"i": 0, "on": False}


def timed(self, a, w, b, c, **kw):
    if not state["on"]:
        return orig(self, a, w, b, c, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    orig(self, a, w, b, c, **kw)
    e.record()
    i = state["i"]
    state["i"] += 1
    if i not in records:
        kh, st, pd = kw.get("kh", 1), kw.get("stride", 1), kw.get("pad", 0)
        ho = (kw["hin"] + 2 * pd - kh) // st + 1
        wo = (kw["win"] + 2 * pd - kh) // st + 1
        m = self.B * ho * wo
        records[i] = dict(name=w, m=m, n=kw["n"], k=kh * kh * kw["cin"], kh=kh, stride=st, hw=f"{kw['hin']}x{kw['win']}",
                          flags=kw.get("split_flags", 0), res=kw.get("res") is not None, c2=kw.get("c2") is not None,
                          mode=kw.get("out_mode", 0), a_elems=self.B * kw["hin"] * kw["win"] * kw["cin"], ev=[])
        order.append(i)
    records[i]["ev"].append((s, e))


ImageTower._gemm = timed
for _ in range(iters):
    state["i"] = 0
    state["on"] = True
    tower.features(imgs)
    state["on"] = False
    torch.cuda.synchronize()

tot = {}
rows = []
for i in order:
    r = records[i]
    us = statistics.median(1e3 * s.elapsed_time(e) for s, e in r["ev"])
    neckhead = not (r["name"].startswith("s") and r["name"][1].isdigit()) and not r["name"].startswith(("down", "stem"))
    if r["name"].startswith("downsample"):
        neckhead = True
    fl = 2.0 * r["m"] * r["n"] * r["k"]
    tf = fl / us / 1e6
    # algorithmic bytes: 4 per element in either format (fp32, or fp16 hi + lo); a dual-format output (+c2) is written twice
    r["bytes"] = 4.0 * (r["a_elems"] + r["n"] * r["k"] + r["m"] * r["n"] * ((2 if r["c2"] else 1) + (1 if r["res"] else 0)))
    r["t_mfma"], r["t_hbm"] = fl / ROOF / 1e6, r["bytes"] / HBM_PEAK * 1e6          # us at each roof
    rows.append((r, us, tf, neckhead))
    key = "neck+head" if neckhead else "backbone"
    t = tot.setdefault(key, [0.0, 0.0])
    t[0] += us
    t[1] += fl
print(f"# {arch} B={B} {S}x{S}, serial chain, median of {iters} steps; us include the launch gap of a dependent chain")
print(f"{'layer':28s} {'map':>9s} {'m':>7s} {'n':>5s} {'k':>5s} {'geom':>6s} {'fl':>3s} {'us':>8s} {'TF':>7s} {'frac':>6s} {'MB':>7s} {'GB/s':>6s} {'roof':>5s} {'of it':>6s}")
for r, us, tf, nh in rows:
    if not nh and not show_all:
        continue
    geom = f"{r['kh']}x{r['kh']}s{r['stride']}" + ("d" if r["mode"] else "")
    bound = "hbm" if r["t_hbm"] > r["t_mfma"] else "mfma"
    print(f"{r['name']:28s} {r['hw']:>9s} {r['m']:7d} {r['n']:5d} {r['k']:5d} {geom:>6s} {r['flags']:3d} {us:8.1f} {tf:7.1f} {tf / ROOF:6.3f}"
          f" {r['bytes'] / 1e6:7.1f} {r['bytes'] / us / 1e3:6.0f} {bound:>5s} {max(r['t_hbm'], r['t_mfma']) / us:6.3f}"
          + (" +res" if r["res"] else "") + (" +c2" if r["c2"] else ""))
for k, (us, fl) in tot.items():
    print(f"# total {k}: {us / 1e3:.3f} ms, {fl / 1e9:.1f} GFLOP, {fl / us / 1e6:.1f} TF = {fl / us / 1e6 / ROOF:.3f} of the fp16x3 roof")
for key, pick in (("neck+head", True), ("backbone", False)):
    sel = [(r, us) for r, us, tf, nh in rows if nh == pick]
    if not sel:
        continue
    hb = [(r, us) for r, us in sel if r["t_hbm"] > r["t_mfma"]]
    mf = [(r, us) for r, us in sel if r["t_hbm"] <= r["t_mfma"]]
    floor = sum(max(r["t_hbm"], r["t_mfma"]) for r, _ in sel)
    print(f"# {key}: {len(hb)} launches whose tighter roof is HBM: {sum(u for _, u in hb) / 1e3:.3f} ms at "
          f"{sum(r['bytes'] for r, _ in hb) / max(1e-9, sum(u for _, u in hb)) / 1e6:.2f} TB/s of 8; {len(mf)} MFMA-bound launches: "
          f"{sum(u for _, u in mf) / 1e3:.3f} ms at {sum(2.0 * r['m'] * r['n'] * r['k'] for r, _ in mf) / max(1e-9, sum(u for _, u in mf)) / 1e6 / ROOF:.3f} "
          f"of the fp16x3 roof; sum of the per-layer tighter roofs {floor / 1e3:.3f} ms")
out = os.environ.get("OUT")
if out:
    json.dump([dict(name=r["name"], m=r["m"], n=r["n"], k=r["k"], kh=r["kh"], stride=r["stride"], us=us, tflops=tf, neck_head=nh)
               for r, us, tf, nh in rows], open(out, "w"), indent=0)
