"""replaces /root/reference/src/utils/util.py for what the inference scripts import from it (scripts/pose2vid.py:26,
scripts/audio2vid.py:27, scripts/vid2vid.py:26): `save_videos_grid`, `save_videos_from_pil` (the output side of the path,
SURVEY.md §8f rank 4), `read_frames`, `get_fps`, `seed_everything` — served by aniportrait_amd.video_io, PyAV imported on first
use, no cv2 / torchvision / einops.  Any other name (`crop_face`, the training scripts' host helpers) is served from the
reference's own module — the next `src/utils/util.py` on the namespace package's path — loaded on first use."""
import importlib.util
import os
import sys

from aniportrait_amd.video_io import get_fps, read_frames, save_videos_from_pil, save_videos_grid, seed_everything  # noqa: F401

_reference = None


def _reference_module():
    global _reference
    if _reference is None:
        import src.utils as pkg
        here = os.path.dirname(os.path.abspath(__file__))
        for d in pkg.__path__:
            cand = os.path.join(d, "util.py")
            if os.path.abspath(d) != here and os.path.isfile(cand):
                spec = importlib.util.spec_from_file_location("src.utils._reference_util", cand)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[spec.name] = mod
                spec.loader.exec_module(mod)
                _reference = mod
                break
        else:
            raise ImportError("src.utils.util: no other src/utils/util.py on the path (the reference checkout) to serve this name")
    return _reference


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    try:
        return getattr(_reference_module(), name)
    except ImportError as e:
        raise AttributeError(f"module 'src.utils.util' has no attribute {name!r} ({e})") from e
