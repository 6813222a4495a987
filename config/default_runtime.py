# Runtime defaults shared by the inference configs of this repository.  Only what the entry scripts read:
# there is no Runner, no hooks and no visualizer on the device path.
default_scope = "mmdet"
log_level = "INFO"
load_from = None
resume = False
backend_args = None
env_cfg = dict(cudnn_benchmark=False, dist_cfg=dict(backend="nccl"))     # "nccl" is RCCL on ROCm
