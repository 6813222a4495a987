"""Build libwedetect_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m wedetect_amd.build [--force]

The library is built IN-TREE (wedetect_amd/libwedetect_hip.so) so that it travels to the
GPU box with the repo snapshot; it is git-ignored.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwedetect_hip.so")
SOURCES = ["abi.hip", "conv_gemm.hip", "split_gemm.hip", "split_gemm_pre.hip", "split_gemm_p8.hip", "split_gemm_p4.hip", "split_gemm_conv.hip", "split_gemm_conv3.hip", "split_gemm_mlp.hip", "split_gemm_mlpw.hip", "probe.hip", "stem.hip", "elementwise.hip", "preprocess.hip", "text.hip", "bricks.hip", "evaluate.hip", "postprocess.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]
for _d in os.environ.get("WD_EXTRA_DEFINES", "").split():   # A/B builds only (e.g. WD_GELU_R3, WD_CSPLIT_PLAIN_STORE): never in a release build
    FLAGS.append("-D" + _d)
if os.environ.get("WD_DEBUG_ABLATIONS") == "1":          # timing-only ablation kernels (wrong results): never in a release build
    FLAGS.append("-DWD_DEBUG_ABLATIONS")


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "wedetect_hip.h")]


def source_hash() -> str:
    """sha256 over the kernel sources and headers the library is built from (sorted names + contents): identifies the
    code a profile was taken on independently of where / when it was compiled (bench.py: traffic_provenance)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for f in files + [os.path.join(ROOT, "include", "wedetect_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


# sources whose kernels issue VMEM instructions from inline asm: their ISA is checked after every compile
# (scripts/check_sgpr_vmem_hazard.py: the gfx9 "VALU writes SGPR -> VMEM reads it" hazard the compiler's recogniser cannot see
# inside asm blocks; scripts/check_asm_loads.py: no instruction may touch a register an asm load is still in flight to)
# kernels whose inline-asm vmcnt counting assumes that the compiler adds no VMEM operation of its own (scratch spills /
# reloads): the build fails if one of them needs scratch (stem.hip)
NO_SCRATCH = {"stem.hip": ["stem_fused_kernel"], "split_gemm_mlpw.hip": ["fused_mlp_wide_kernel"],
              "split_gemm_mlp.hip": ["fused_mlp128_kernel"], "split_gemm_p8.hip": ["split_gemm_p8_kernel"],
              "split_gemm_p4.hip": ["split_gemm_p4_kernel"], "split_gemm_conv.hip": ["split_conv_pp_kernel"],
              "split_gemm_conv3.hip": ["split_conv3_kernel", "split_conv3w_kernel"],
              "split_gemm_pre.hip": ["split_gemm_pingpong_kernel", "split_gemm_glds_kernel"],
              "elementwise.hip": ["dwconv7_dma_kernel", "dwconv7_ln_reg4_dma_kernel"]}
ASM_VMEM_SOURCES = {"split_gemm_mlpw.hip": ["fused_mlp_wide_kernel"], "split_gemm_mlp.hip": [], "split_gemm_p8.hip": [],
                    "split_gemm_p4.hip": [], "split_gemm_pre.hip": [], "split_gemm_conv.hip": [], "split_gemm_conv3.hip": [], "stem.hip": [], "elementwise.hip": []}


# kernels that issue ds_reads from inline asm (invisible to the compiler's waitcnt pass): scripts/check_asm_ds_reads.py
ASM_DS_READ_SOURCES = {"split_gemm_conv3.hip": ["split_conv3w_kernel"]}


def check_isa(src: str, verbose: bool = True) -> None:
    """hipcc -S of one source + the two static checks; raises on a hazard."""
    asm = _obj(src).replace(".o", ".s")
    cmd = [HIPCC, *[f for f in FLAGS if f != "-fPIC"], "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", asm]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    scripts = os.path.join(ROOT, "scripts")
    r = subprocess.run([sys.executable, os.path.join(scripts, "check_sgpr_vmem_hazard.py"), asm], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"SGPR -> VMEM hazard in the ISA of {src}:\n{r.stdout[-2000:]}")
    syms = ASM_VMEM_SOURCES[src]
    if syms:
        r = subprocess.run([sys.executable, os.path.join(scripts, "check_asm_loads.py"), asm, *syms], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"pending asm load touched in the ISA of {src}:\n{r.stdout[-2000:]}")
    if src in ASM_DS_READ_SOURCES:
        r = subprocess.run([sys.executable, os.path.join(scripts, "check_asm_ds_reads.py"), asm, *ASM_DS_READ_SOURCES[src]], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"pending asm ds_read touched in the ISA of {src}:\n{r.stdout[-2000:]}")
    if src in NO_SCRATCH:
        import re
        meta = open(asm).read()
        meta = meta[meta.index("amdhsa.kernels"):] if "amdhsa.kernels" in meta else ""
        seen = set()
        for blk in meta.split("- .agpr_count")[1:]:               # one metadata record per kernel
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            for k in NO_SCRATCH[src]:
                if k in name:
                    seen.add(k)
                    if scratch != 0:
                        raise RuntimeError(f"{src}: kernel {name} uses {scratch} bytes of scratch — its counted vmcnt waits assume none")
        missing = [k for k in NO_SCRATCH[src] if k not in seen]
        if missing:
            raise RuntimeError(f"{src}: NO_SCRATCH names no kernel of this source: {missing}")
    os.remove(asm)
    if verbose:
        print(f"ISA checks passed: {src}", flush=True)


def _obj(src: str) -> str:
    return os.path.join(CSRC, src.replace(".hip", ".o"))


def _obj_stale(src: str) -> bool:
    obj = _obj(src)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + _headers())


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_obj_stale(s) or os.path.getmtime(_obj(s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = True) -> str:
    """Recompiles the sources whose object is older than the source or any header (all of them in a fresh
    checkout: objects are not tracked), in parallel, then links."""
    if not force and not _stale():
        return LIB
    procs = []
    for src in SOURCES:
        if not force and not _obj_stale(src):
            continue
        cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=4) as ex:                 # the ISA checks of the rebuilt asm-VMEM sources run beside the compiles
        checks = [ex.submit(check_isa, src, verbose) for src, _ in procs if src in ASM_VMEM_SOURCES]
        for src, p in procs:
            if p.wait() != 0:
                raise RuntimeError(f"hipcc failed on {src}")
        for c in checks:
            c.result()
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
