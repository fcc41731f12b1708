"""replaces /root/reference/src/models/mutual_self_attention.py"""
from aniportrait_amd.mutual_self_attention import ReferenceAttentionControl  # noqa: F401
