"""Test-only stub of `diffusers.image_processor.VaeImageProcessor.preprocess` (0.24.0
restated).  PIL path: RGB -> lanczos resize -> /255 -> NCHW -> 2x-1.  numpy path (what the
scripts pass for pose images, `scripts/pose2vid.py:158,168`): stack -> NCHW tensor (dtype
kept, i.e. uint8, NO /255) -> F.interpolate(nearest) -> 2x-1 unless min < 0."""
import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class VaeImageProcessor:
    def __init__(self, do_resize=True, vae_scale_factor=8, resample="lanczos", do_normalize=True,
                 do_binarize=False, do_convert_rgb=False, do_convert_grayscale=False):
        self.config = _Cfg(do_resize=do_resize, vae_scale_factor=vae_scale_factor, resample=resample,
                           do_normalize=do_normalize, do_convert_rgb=do_convert_rgb)

    def _hw(self, image, height, width):
        if height is None:
            height = image.height if isinstance(image, PIL.Image.Image) else image.shape[-2 if torch.is_tensor(image) else 1]
        if width is None:
            width = image.width if isinstance(image, PIL.Image.Image) else image.shape[-1 if torch.is_tensor(image) else 2]
        f = self.config.vae_scale_factor
        return height - height % f, width - width % f

    def preprocess(self, image, height=None, width=None):
        if isinstance(image, (PIL.Image.Image, np.ndarray, torch.Tensor)):
            image = [image]
        if isinstance(image[0], PIL.Image.Image):
            if self.config.do_convert_rgb:
                image = [i.convert("RGB") for i in image]
            if self.config.do_resize:
                h, w = self._hw(image[0], height, width)
                image = [i.resize((w, h), resample=PIL.Image.LANCZOS) for i in image]
            arr = np.stack([np.array(i).astype(np.float32) / 255.0 for i in image], axis=0)
            image = torch.from_numpy(arr.transpose(0, 3, 1, 2))
        elif isinstance(image[0], np.ndarray):
            arr = np.concatenate(image, axis=0) if image[0].ndim == 4 else np.stack(image, axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            image = torch.from_numpy(arr.transpose(0, 3, 1, 2))
            h, w = self._hw(image, height, width)
            if self.config.do_resize:
                image = F.interpolate(image, size=(h, w))
        else:
            raise NotImplementedError("tensor inputs are not used by the reference scripts")
        do_normalize = self.config.do_normalize
        if image.min() < 0 and do_normalize:
            do_normalize = False
        if do_normalize:
            image = 2.0 * image - 1.0
        return image
