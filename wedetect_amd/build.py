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
SOURCES = ["abi.hip", "conv_gemm.hip", "split_gemm.hip", "split_gemm_pre.hip", "split_gemm_p8.hip", "split_gemm_p4.hip", "split_gemm_conv.hip", "split_gemm_mlp.hip", "split_gemm_mlpw.hip", "probe.hip", "stem.hip", "elementwise.hip", "preprocess.hip", "text.hip", "bricks.hip", "evaluate.hip", "postprocess.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]
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
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
