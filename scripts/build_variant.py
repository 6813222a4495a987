"""A second build of the library with extra -D defines, for same-box A/B runs through $WEDETECT_LIB:
    python scripts/build_variant.py gelu_r4 WD_GELU_R4      ->  wedetect_amd/libwedetect_hip_gelu_r4.so
Objects go to /tmp (nothing of it is tracked; the .so travels with the gpurun snapshot like the release build)."""
import concurrent.futures as cf
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wedetect_amd import build as wb

name, defines = sys.argv[1], sys.argv[2:]
odir = os.path.join("/tmp", "wd_variant_" + name)
os.makedirs(odir, exist_ok=True)
flags = wb.FLAGS + ["-D" + d for d in defines]


def cc(src):
    obj = os.path.join(odir, src.replace(".hip", ".o"))
    subprocess.check_call([wb.HIPCC, *flags, "-c", os.path.join(wb.CSRC, src), "-o", obj], stderr=subprocess.DEVNULL)
    return obj


with cf.ThreadPoolExecutor(max_workers=8) as ex:
    objs = list(ex.map(cc, wb.SOURCES))
out = os.path.join(wb.HERE, f"libwedetect_hip_{name}.so")
subprocess.check_call([wb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
print(out)
