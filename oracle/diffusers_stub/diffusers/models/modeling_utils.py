"""Test-only stub of `diffusers.models.modeling_utils.ModelMixin` (0.24.0 behaviour restated):
an `nn.Module` with `.dtype/.device`, config attribute fall-through and a minimal
`from_pretrained` (config.json + diffusion_pytorch_model.{safetensors,bin}, `.eval()`)."""
import os

import torch
import torch.nn as nn


class ModelMixin(nn.Module):
    config_name = "config.json"
    _supports_gradient_checkpointing = False

    def __init__(self):
        super().__init__()

    def __getattr__(self, name):
        d = self.__dict__.get("_internal_dict", None)
        if d is not None and name in d and name not in self.__dict__:
            return d[name]
        return super().__getattr__(name)

    @property
    def dtype(self):
        for p in self.parameters():
            return p.dtype
        for b in self.buffers():
            return b.dtype
        return torch.float32

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kwargs):
        from safetensors.torch import load_file

        from ..utils import SAFETENSORS_WEIGHTS_NAME, WEIGHTS_NAME

        path = str(path)
        if subfolder is not None:
            path = os.path.join(path, subfolder)
        config = cls.load_config(os.path.join(path, cls.config_name))
        model = cls.from_config(config)
        st = os.path.join(path, SAFETENSORS_WEIGHTS_NAME)
        if os.path.exists(st):
            sd = load_file(st, device="cpu")
        else:
            sd = torch.load(os.path.join(path, WEIGHTS_NAME), map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=False)
        model.eval()
        return model
