"""replaces /root/reference/src/pipelines/context.py"""
from aniportrait_amd.context import get_context_scheduler, ordered_halving, uniform  # noqa: F401
