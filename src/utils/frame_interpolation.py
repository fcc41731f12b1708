"""replaces /root/reference/src/utils/frame_interpolation.py (scripts/pose2vid.py:27,124,178-179: the `-acc` path)"""
from aniportrait_amd.frame_interpolation import batch_images_interpolation_tool, init_frame_interpolation_model  # noqa: F401
