"""`VaeImageProcessor.preprocess` and `randn_tensor` with diffusers 0.24.0 semantics for the inputs
the scripts pass (`src/pipelines/pipeline_pose2vid_long.py:75-83,149-183,424-452`).

* PIL image (reference portrait): RGB -> lanczos resize to multiples of 8 -> [0,1] -> NCHW -> 2x-1.
* numpy uint8 (H,W,3) pose renderings (`scripts/pose2vid.py:158,168`): NCHW tensor with the dtype
  KEPT (no /255), nearest resize, then 2x-1 — values in [-1, 509].  Harmless downstream because the
  PoseGuider's first conv is followed by train-mode BatchNorm; replicated.
"""
import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F


class VaeImageProcessor:
    def __init__(self, do_resize=True, vae_scale_factor=8, resample="lanczos", do_normalize=True,
                 do_binarize=False, do_convert_rgb=False, do_convert_grayscale=False):
        if resample != "lanczos" or do_binarize or do_convert_grayscale:
            raise NotImplementedError("VaeImageProcessor: only the options the pose2vid pipelines use")
        self.do_resize, self.scale, self.do_normalize, self.to_rgb = do_resize, vae_scale_factor, do_normalize, do_convert_rgb

    def _target(self, h0, w0, height, width):
        h = h0 if height is None else height
        w = w0 if width is None else width
        return h - h % self.scale, w - w % self.scale

    def preprocess(self, image, height=None, width=None):
        items = image if isinstance(image, (list, tuple)) else [image]
        first = items[0]
        if isinstance(first, PIL.Image.Image):
            frames = []
            for im in items:
                if self.to_rgb:
                    im = im.convert("RGB")
                if self.do_resize:
                    h, w = self._target(im.height, im.width, height, width)
                    im = im.resize((w, h), resample=PIL.Image.LANCZOS)
                frames.append(np.asarray(im, dtype=np.float32) / 255.0)
            t = torch.from_numpy(np.stack(frames).transpose(0, 3, 1, 2).copy())
        elif isinstance(first, np.ndarray):
            arr = np.concatenate(items, axis=0) if first.ndim == 4 else np.stack(items, axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            t = torch.from_numpy(np.ascontiguousarray(arr.transpose(0, 3, 1, 2)))
            if self.do_resize:
                h, w = self._target(t.shape[2], t.shape[3], height, width)
                t = F.interpolate(t, size=(h, w))
        else:
            raise NotImplementedError("VaeImageProcessor.preprocess: PIL images or numpy arrays only")
        if self.do_normalize and not bool(t.min() < 0):
            t = 2.0 * t - 1.0
        return t


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """A CPU generator samples on the CPU (in `dtype`) and the result is moved: initial latents are
    vendor-independent (SURVEY.md §5 RNG)."""
    target = torch.device(device) if device is not None else torch.device("cpu")
    gens = generator if isinstance(generator, (list, tuple)) else [generator]
    on = target
    if gens[0] is not None and gens[0].device.type == "cpu" and target.type != "cpu":
        on = torch.device("cpu")
    if isinstance(generator, (list, tuple)):
        parts = [torch.randn((1,) + tuple(shape[1:]), generator=g, device=on, dtype=dtype) for g in gens]
        return torch.cat(parts, dim=0).to(target)
    return torch.randn(tuple(shape), generator=generator, device=on, dtype=dtype).to(target)
