"""replaces /root/reference/src/models/pose_guider.py"""
from aniportrait_amd.pose_guider import PoseGuider  # noqa: F401
