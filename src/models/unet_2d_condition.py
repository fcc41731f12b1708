"""replaces /root/reference/src/models/unet_2d_condition.py"""
from aniportrait_amd.unet import UNet2DConditionModel, UNet2DConditionOutput  # noqa: F401
