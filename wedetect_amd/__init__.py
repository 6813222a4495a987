"""wedetect_amd — MI355X-native (gfx950) implementation of WeDetect's dual-tower
inference hot path: ConvNeXt image tower + CSPRepBiFPAN neck + YOLO-World head ->
region embeddings -> region x text similarity -> sort / top-k / class-aware NMS,
plus the 8-GPU region-embedding gather used for object retrieval.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all compute goes through the C-ABI library ``libwedetect_hip.so`` built from
``wedetect_amd/csrc`` (declared in ``include/wedetect_hip.h``).  There is no CPU
fallback in this package: if the library is missing, importing ``wedetect_amd.lib``
raises.  The CPU restatement used as the parity checker lives in ``/oracle`` and is
never imported from here.
"""

__version__ = "0.1.0"

from .arch import ARCHS, ArchSpec, get_arch  # noqa: F401
