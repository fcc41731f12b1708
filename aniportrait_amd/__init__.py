"""aniportrait_amd — MI355X-native (gfx950) implementation of AniPortrait's pose2vid frame-batch denoising
hot path behind the reference's own Python operator API.  See DESIGN.md / INTEGRATION.md.

Importing the package never touches the GPU or the shared library; the first compute call does, and
fails loudly if `aniportrait_amd/lib/libaniportrait_hip.so` is missing (build: `python -m aniportrait_amd.build`).
"""
__version__ = "0.1.0"
