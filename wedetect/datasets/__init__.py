"""``wedetect.datasets``: the test-pipeline transforms (device versions)."""
from wedetect_amd.pipeline import (Compose, LoadAnnotations, LoadImageFromFile, LoadText, PackDetInputs,  # noqa: F401
                                   WeDetectKeepRatioResize, WeDetectLetterResize)

__all__ = ["Compose", "LoadImageFromFile", "WeDetectKeepRatioResize", "WeDetectLetterResize", "LoadAnnotations", "LoadText",
           "PackDetInputs"]
