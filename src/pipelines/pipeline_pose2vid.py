"""replaces /root/reference/src/pipelines/pipeline_pose2vid.py"""
from aniportrait_amd.pipeline_pose2vid import Pose2VideoPipeline, Pose2VideoPipelineOutput  # noqa: F401
