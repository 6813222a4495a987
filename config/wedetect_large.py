# WeDetect-Large: Large sizes, XLM-R-large text tower, 1280 x 1280 inputs.
_base_ = ["wedetect_tiny.py"]

size = "large"
img_scale = (1280, 1280)
model = dict(
    backbone=dict(image_model=dict(model_name=size),
                  text_model=dict(model_name="./xlm-roberta-large/", model_size=size)),
    neck=dict(scale_factor=1.5, model_size=size),
    bbox_head=dict(head_module=dict(model_size=size)))
test_pipeline = [
    dict(type="LoadImageFromFile", backend_args=None),
    dict(type="WeDetectKeepRatioResize", scale=img_scale),
    dict(type="WeDetectLetterResize", scale=img_scale, allow_scale_up=False, pad_val=dict(img=114)),
    dict(type="LoadAnnotations", with_bbox=True, _scope_="mmdet"),
    dict(type="LoadText"),
    dict(type="PackDetInputs",
         meta_keys=("img_id", "img_path", "ori_shape", "img_shape", "scale_factor", "pad_param", "texts")),
]
