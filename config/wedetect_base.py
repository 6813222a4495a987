# WeDetect-Base: the Tiny inference config with the Base sizes.
_base_ = ["wedetect_tiny.py"]

size = "base"
model = dict(
    backbone=dict(image_model=dict(model_name=size), text_model=dict(model_size=size)),
    neck=dict(scale_factor=1.0, model_size=size),
    bbox_head=dict(head_module=dict(model_size=size)))
