# WeDetect-Tiny, inference view.  Evaluates to the same `model`, `img_scale` and `test_pipeline` as the reference's
# config/wedetect_tiny.py (checked by tests/test_cpu.py against tests/golden/model_cfgs.json); the training,
# dataset and evaluator sections of that file are outside this repository's scope, and a reference config file can
# be used here unchanged (wedetect_amd.cfgfile.Config reads it).
_base_ = ["default_runtime.py"]

size = "tiny"
text_dir = "./xlm-roberta-base/"
num_classes = 1203                 # LVIS vocabulary (test)
num_training_classes = 80
text_channels = 768

custom_imports = dict(imports=["wedetect"], allow_failed_imports=False)

model_test_cfg = dict(multi_label=True, nms_pre=30000, score_thr=0.001, nms=dict(type="nms", iou_threshold=0.7),
                      max_per_img=300)

_neck = dict(type="CSPRepBiFPANNeck", model_size=size)

model = dict(
    type="YOLOWorldDetector",
    mm_neck=False,
    num_train_classes=num_training_classes,
    num_test_classes=num_classes,
    data_preprocessor=dict(type="YOLOWDetDataPreprocessor", mean=[0.0, 0.0, 0.0], std=[255.0, 255.0, 255.0], bgr_to_rgb=True),
    backbone=dict(
        type="MultiModalYOLOBackbone",
        image_model=dict(type="ConvNextVisionBackbone", model_name=size, frozen_modules=[]),
        text_model=dict(type="XLMRobertaLanguageBackbone", model_name=text_dir, model_size=size, frozen_modules=[])),
    neck=_neck,
    bbox_head=dict(
        type="YOLOWorldHead",
        head_module=dict(type="YOLOWorldHeadModule", use_bn_head=True, embed_dims=text_channels,
                         num_classes=num_training_classes, model_size=size, in_channels=[256, 512, 1024]),
        prior_generator=dict(type="MlvlPointGenerator", offset=0.5, strides=[8, 16, 32]),
        bbox_coder=dict(type="WeDetectDistancePointBBoxCoder"),
        loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=True, reduction="none", loss_weight=0.5),
        loss_bbox=dict(type="mmyoloIoULoss", iou_mode="ciou", bbox_format="xyxy", reduction="sum", loss_weight=7.5,
                       return_iou=False),
        loss_dfl=dict(type="DistributionFocalLoss", reduction="mean", loss_weight=1.5 / 4)),
    train_cfg=dict(assigner=dict(type="BatchTaskAlignedAssigner", num_classes=num_classes, use_ciou=True, topk=10,
                                 alpha=0.5, beta=6.0, eps=1e-9)),
    test_cfg=model_test_cfg)

img_scale = (640, 640)             # (w, h)

test_pipeline = [
    dict(type="LoadImageFromFile", backend_args=None),
    dict(type="WeDetectKeepRatioResize", scale=img_scale),
    dict(type="WeDetectLetterResize", scale=img_scale, allow_scale_up=False, pad_val=dict(img=114)),
    dict(type="LoadAnnotations", with_bbox=True, _scope_="mmdet"),
    dict(type="LoadText"),
    dict(type="PackDetInputs",
         meta_keys=("img_id", "img_path", "ori_shape", "img_shape", "scale_factor", "pad_param", "texts")),
]
