"""replaces /root/reference/src/pipelines/pipeline_pose2vid_long.py"""
from aniportrait_amd.pipeline_pose2vid_long import Pose2VideoPipeline, Pose2VideoPipelineOutput  # noqa: F401
