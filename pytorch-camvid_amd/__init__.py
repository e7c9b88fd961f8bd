"""pytorch_camvid_amd — MI355X-native (gfx950) execution of the pytorch-camvid UNet/SegNet training hot path.

Import name: `pytorch_camvid_amd` (the directory is `pytorch-camvid_amd/`; `pytorch_camvid_amd.py` at the repository
root aliases it).  Public surface mirrors the reference (models/unet.py, models/segnet.py, utils.get_model,
train.py's loss): UNet, SegNet, BasicConv2d, BasicConv, UpSample2d, get_model, CrossEntropyLoss.
"""
from ._lib import CvkError, build as build_library, load as load_library          # noqa: F401
from .modules import BasicConv, BasicConv2d, SegNet, UNet, UpSample2d, get_model, set_conv_precision, set_split_operands   # noqa: F401
from .functional import (ConfusionMeter, CrossEntropyLoss, argmax_channels, cross_entropy, evaluate,  # noqa: F401
                         evaluate_report, predict, preprocess_uint8, DevicePrefetcher, last_ce_status)
from .optim import FlatAdamW  # noqa: F401
from . import ddp  # noqa: F401
from .graph import GraphedStep  # noqa: F401
from .engine import mark_weights_dirty  # noqa: F401
from .checkpoint import (save_checkpoint, load_checkpoint, latest_checkpoint, checkpoint_epoch, resume,   # noqa: F401
                         save_policy, reference_state_dict)

__all__ = ["UNet", "SegNet", "BasicConv2d", "BasicConv", "UpSample2d", "get_model", "set_conv_precision", "set_split_operands", "CrossEntropyLoss",
           "cross_entropy", "last_ce_status", "argmax_channels", "ConfusionMeter", "evaluate", "evaluate_report", "predict", "preprocess_uint8", "DevicePrefetcher", "FlatAdamW", "ddp", "GraphedStep", "mark_weights_dirty", "save_checkpoint", "load_checkpoint", "latest_checkpoint", "checkpoint_epoch", "resume", "save_policy",
           "reference_state_dict", "build_library", "load_library", "CvkError"]
