"""Output side: aniportrait_amd.video_io.save_videos_grid against a plain restatement of what the
reference does per frame (/root/reference/src/utils/util.py:87-104 on top of torchvision.utils.make_grid: padding 2,
`nrow` cells per row, a single image returned as it is), for the script's 3-column grid, a ragged last row, one video,
`rescale`, and for the display bytes the pipeline makes on the device (`output_type="uint8"`)."""
import numpy as np
import pytest
import torch

from aniportrait_amd import video_io


def _reference_frames(videos, rescale, n_rows):
    """per time step: grid (zeros, cells at r (h + 2) + 2 / c (w + 2) + 2) -> (h, w, c) -> optional (x + 1) / 2 ->
    uint8(x * 255); one image: no border"""
    b, c, t, h, w = videos.shape
    out = []
    for i in range(t):
        x = videos[:, :, i]
        if b == 1:
            g = x[0]
        else:
            cols = min(n_rows, b)
            rows = (b + cols - 1) // cols
            g = torch.zeros(c, rows * (h + 2) + 2, cols * (w + 2) + 2)
            for k in range(b):
                y0, x0 = (k // cols) * (h + 2) + 2, (k % cols) * (w + 2) + 2
                g[:, y0:y0 + h, x0:x0 + w] = x[k]
        g = g.permute(1, 2, 0)
        if rescale:
            g = (g + 1.0) / 2.0
        out.append((g * 255).numpy().astype(np.uint8))
    return np.stack(out)


def _capture(monkeypatch):
    got = {}

    def fake_save(pil_images, path, fps=8):
        got["frames"] = np.stack([np.asarray(im) for im in pil_images])
        got["path"], got["fps"] = path, fps
    monkeypatch.setattr(video_io, "save_videos_from_pil", fake_save)
    return got


@pytest.mark.parametrize("b,n_rows,rescale", [(3, 3, False), (7, 3, False), (1, 6, False), (4, 6, True), (2, 1, False)])
def test_grid_frames_equal_the_reference_algorithm(monkeypatch, b, n_rows, rescale):
    g = torch.Generator().manual_seed(7 + b)
    v = torch.rand((b, 3, 5, 12, 10), generator=g)
    v[0, :, 0, 0, :4] = torch.tensor([0.0, 1.0, 254.999 / 255, 0.5])[None]      # exact ends and a truncation case
    if rescale:
        v = v * 2 - 1
    got = _capture(monkeypatch)
    video_io.save_videos_grid(v, "out/x.mp4", rescale=rescale, n_rows=n_rows, fps=12)
    want = _reference_frames(v, rescale, n_rows)
    assert got["frames"].shape == want.shape and got["frames"].dtype == np.uint8
    assert np.array_equal(got["frames"], want)
    assert got["path"] == "out/x.mp4" and got["fps"] == 12


def test_display_bytes_from_the_pipeline_give_the_same_file_frames(monkeypatch):
    """the script's call: cat([ref, pose, video]) fp32 -> save; and the same three clips as uint8 (b, t, h, w, 3)"""
    g = torch.Generator().manual_seed(3)
    video = torch.rand((1, 3, 4, 16, 16), generator=g).half().float()      # what decode_latents hands over: fp16 values
    ref, pose = torch.rand((1, 3, 4, 16, 16), generator=g), torch.rand((1, 3, 4, 16, 16), generator=g)
    cat = torch.cat([ref, pose, video], dim=0)
    got = _capture(monkeypatch)
    video_io.save_videos_grid(cat, "a.gif", n_rows=3, fps=8)
    floats = got["frames"].copy()
    u8 = video_io.display_bytes(cat)
    assert u8.shape == (3, 4, 16, 16, 3) and u8.dtype == torch.uint8
    video_io.save_videos_grid(u8, "a.gif", n_rows=3, fps=8)
    assert np.array_equal(got["frames"], floats)
    video_io.save_videos_grid(u8[2].numpy(), "a.gif")                        # one clip as (t, h, w, 3): no border
    assert np.array_equal(got["frames"], u8[2].numpy())
    with pytest.raises(ValueError):
        video_io.save_videos_grid(u8, "a.gif", rescale=True)

    class Pending:                                                           # async_output=True hands over a future
        def result(self):
            return u8
    video_io.save_videos_grid(Pending(), "a.gif", n_rows=3)
    assert np.array_equal(got["frames"], floats)


def test_gif_file_and_unsupported_suffix(tmp_path):
    from PIL import Image
    v = torch.rand((2, 3, 3, 8, 8), generator=torch.Generator().manual_seed(1))
    p = tmp_path / "sub" / "clip.gif"
    video_io.save_videos_grid(v, str(p), n_rows=2, fps=5)
    im = Image.open(p)
    assert im.n_frames == 3 and im.size == (2 * 10 + 2, 12)
    with pytest.raises(ValueError):
        video_io.save_videos_grid(v, str(tmp_path / "clip.avi"))
    try:
        import av  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="PyAV"):
            video_io.save_videos_grid(v, str(tmp_path / "clip.mp4"))


def test_script_import_block_resolves_without_the_reference_util(tmp_path):
    """`from src.utils.util import get_fps, read_frames, save_videos_grid` (scripts/pose2vid.py:26) and `seed_everything`
    come from this repository: no cv2 / torchvision / einops import, PyAV only when a video is opened"""
    import random
    import sys

    from src.utils import util as shim
    for name in ("get_fps", "read_frames", "save_videos_grid", "save_videos_from_pil", "seed_everything"):
        assert getattr(shim, name) is getattr(video_io, name)
    assert "src.utils._reference_util" not in sys.modules or shim._reference is not None    # nothing forced the fall-through
    shim.seed_everything(123)
    a = (random.random(), float(np.random.rand()), float(torch.rand(())))
    shim.seed_everything(123)
    assert a == (random.random(), float(np.random.rand()), float(torch.rand(())))
    try:
        import av  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="PyAV"):
            shim.read_frames(str(tmp_path / "pose.mp4"))
        with pytest.raises(ImportError, match="PyAV"):
            shim.get_fps(str(tmp_path / "pose.mp4"))
