"""replaces /root/reference/src/models/unet_3d.py"""
from aniportrait_amd.unet import UNet3DConditionModel, UNet3DConditionOutput  # noqa: F401
