"""Output side of the path: what the reference's scripts do with the frames the pipeline returns —
`save_videos_grid` and `save_videos_from_pil` (/root/reference/src/utils/util.py:51-104) — and the three small helpers every
inference script imports next to them for the INPUT side, `read_frames`, `get_fps` (the pose video, util.py:107-130) and
`seed_everything` (util.py:17-25), so that `from src.utils.util import ...` at the top of scripts/pose2vid.py / vid2vid.py /
audio2vid.py resolves without executing the reference's util.py (which imports cv2, torchvision and einops at module level).
PyAV is imported on first use.  What is left to the reference's own file through the `src/utils/util.py` shim: the training
scripts' host helpers (`import_filename`, `delete_additional_ckpt`) and `crop_face` — outside the path's scope.

The reference builds every output frame on the host in fp32: `torchvision.utils.make_grid` of the (b, 3, h, w) batch of
one time step, two transposes, `(x * 255).numpy().astype(uint8)`, `Image.fromarray` — per frame, in a Python loop
(util.py:92-100).  Here the video is turned into display bytes ONCE (on whatever device it lives on; the pipeline's
`output_type="uint8"` hands them over ready-made, a quarter of the D2H bytes) and the grid of every time step is pasted
in uint8 with one vectorised copy per grid cell.  The frames are the reference's, byte for byte:
  * cell (r, c) of the grid starts at row r (h + 2) + 2, column c (w + 2) + 2 (make_grid: padding 2, cells per row
    = min(n_rows, b), rows = ceil(b / cells per row)); a batch of ONE video is returned as it is, without a border;
  * border pixels are 0 — after `rescale` ((x + 1) / 2 applied to the whole grid, util.py:95-96) they are 0.5 -> 127;
  * the conversion truncates: uint8(x * 255) in fp32.
"""
import os

import numpy as np
import torch


def _av():
    try:
        import av
    except ImportError as e:  # pragma: no cover - depends on the installation
        raise ImportError("PyAV (`av`) is required for .mp4 output and for reading pose videos, as in the reference "
                          "(requirements.txt: av==11.0.0)") from e
    return av


def display_bytes(videos, rescale=False):
    """float video (b, 3, t, h, w) in [0, 1] ([-1, 1] with rescale) -> uint8 (b, t, h, w, 3), the bytes the reference
    writes: uint8(x * 255) in fp32, truncating (util.py:95-97).  Runs on the tensor's device.  Precondition: the values
    are inside the stated range (the pipeline clamps: decode_latents); a float outside [0, 256) converts to an
    implementation-defined byte on the device as it does in numpy, and the two need not agree there."""
    x = videos.float()
    if rescale:
        x = (x + 1.0) / 2.0
    return (x * 255).to(torch.uint8).permute(0, 2, 3, 4, 1).contiguous()


def grid_frames(frames_u8, n_rows=6, border=0):
    """uint8 (b, t, h, w, 3) -> uint8 (t, H, W, 3): the make_grid layout described in the module docstring"""
    f = np.asarray(frames_u8.cpu() if isinstance(frames_u8, torch.Tensor) else frames_u8)
    assert f.ndim == 5 and f.shape[-1] == 3 and f.dtype == np.uint8, f"expected uint8 (b, t, h, w, 3), got {f.dtype} {f.shape}"
    b, t, h, w, _ = f.shape
    if b == 1:
        return np.ascontiguousarray(f[0])
    cols = min(int(n_rows), b)
    rows = -(-b // cols)
    out = np.full((t, rows * (h + 2) + 2, cols * (w + 2) + 2, 3), border, dtype=np.uint8)
    for k in range(b):
        y0, x0 = (k // cols) * (h + 2) + 2, (k % cols) * (w + 2) + 2
        out[:, y0:y0 + h, x0:x0 + w] = f[k]
    return out


def save_videos_from_pil(pil_images, path, fps=8):
    """util.py:51-84: .mp4 (libx264 through PyAV) or .gif (PIL), anything else is a ValueError"""
    ext = os.path.splitext(path)[1]
    if os.path.dirname(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
    if ext == ".mp4":
        av = _av()
        width, height = pil_images[0].size
        container = av.open(path, "w")
        stream = container.add_stream("libx264", rate=fps)
        stream.width, stream.height = width, height
        for im in pil_images:
            container.mux(stream.encode(av.VideoFrame.from_image(im)))
        container.mux(stream.encode())
        container.close()
    elif ext == ".gif":
        pil_images[0].save(fp=path, format="GIF", append_images=pil_images[1:], save_all=True, duration=(1 / fps * 1000),
                           loop=0)
    else:
        raise ValueError("Unsupported file type. Use .mp4 or .gif.")


def save_videos_grid(videos, path, rescale=False, n_rows=6, fps=8):
    """util.py:87-104, same signature.  `videos`: the reference's float tensor (b, 3, t, h, w) — e.g. the script's
    `torch.cat([ref_image_tensor, pose_tensor, video], dim=0)` (scripts/pose2vid.py:160-161) — on any device; or display
    bytes, uint8 (b, t, h, w, 3) / (t, h, w, 3) as the pipeline returns them with `output_type="uint8"` (`rescale` must
    then be False: the bytes are final); or a pending asynchronous result (`.result()` is taken)."""
    from PIL import Image
    if hasattr(videos, "result") and not isinstance(videos, torch.Tensor):
        videos = videos.result()
    if isinstance(videos, np.ndarray):
        videos = torch.from_numpy(videos)
    if videos.dtype == torch.uint8:
        if rescale:
            raise ValueError("save_videos_grid: rescale applies to float videos; display bytes are final")
        u8 = videos if videos.dim() == 5 else videos[None]
    else:
        if videos.dim() != 5:
            raise ValueError(f"save_videos_grid: expected a (b, c, t, h, w) video, got {tuple(videos.shape)}")
        u8 = display_bytes(videos, rescale)
    frames = grid_frames(u8, n_rows, border=127 if rescale else 0)
    save_videos_from_pil([Image.fromarray(fr) for fr in frames], path, fps)


def read_frames(video_path):
    """every frame of the first video stream as an RGB PIL image (util.py:107-121; scripts/pose2vid.py:128 reads the pose
    video with it)"""
    av = _av()
    with av.open(video_path) as container:
        stream = next(st for st in container.streams if st.type == "video")
        return [frame.to_image().convert("RGB") for frame in container.decode(stream)]


def get_fps(video_path):
    """average frame rate of the first video stream, as PyAV reports it — a Fraction (util.py:124-129)"""
    av = _av()
    with av.open(video_path) as container:
        return next(st for st in container.streams if st.type == "video").average_rate


def seed_everything(seed):
    """torch (CPU + every GPU), numpy and `random` from one seed (util.py:17-25)"""
    import random
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
