"""``wedetect.models``: the registry names of the reference's model package that are on the inference path."""
from wedetect_amd.bricks import ImagePoolingAttentionModule, MaxSigmoidAttnBlock  # noqa: F401
from wedetect_amd.config import (ConvNextVisionBackbone, CSPRepBiFPANNeck, MlvlPointGenerator,  # noqa: F401
                                 WeDetectDistancePointBBoxCoder, YOLOWDetDataPreprocessor, YOLOWorldHead,
                                 YOLOWorldHeadModule)
from wedetect_amd.detector import MultiModalYOLOBackbone, YOLOWorldDetector  # noqa: F401
from wedetect_amd.text import XLMRobertaLanguageBackbone  # noqa: F401

__all__ = ["YOLOWorldDetector", "MultiModalYOLOBackbone", "ConvNextVisionBackbone", "XLMRobertaLanguageBackbone",
           "CSPRepBiFPANNeck", "YOLOWorldHead", "YOLOWorldHeadModule", "YOLOWDetDataPreprocessor", "MlvlPointGenerator",
           "WeDetectDistancePointBBoxCoder", "MaxSigmoidAttnBlock", "ImagePoolingAttentionModule"]
