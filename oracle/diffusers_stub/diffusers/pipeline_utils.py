"""Test-only stub of `diffusers.DiffusionPipeline`: register_modules / to / device /
progress_bar — all the reference pipeline uses (`src/pipelines/pipeline_pose2vid_long.py:58-70,
368,458`)."""
import torch
import torch.nn as nn
from tqdm.auto import tqdm


class DiffusionPipeline:
    def register_modules(self, **kwargs):
        self._modules_registered = getattr(self, "_modules_registered", [])
        for k, v in kwargs.items():
            setattr(self, k, v)
            self._modules_registered.append(k)

    def to(self, device=None, dtype=None):
        for k in getattr(self, "_modules_registered", []):
            m = getattr(self, k)
            if isinstance(m, nn.Module):
                if dtype is not None:
                    m.to(device=device, dtype=dtype)
                else:
                    m.to(device=device)
        return self

    @property
    def device(self):
        for k in getattr(self, "_modules_registered", []):
            m = getattr(self, k)
            if isinstance(m, nn.Module):
                for p in m.parameters():
                    return p.device
        return torch.device("cpu")

    def progress_bar(self, iterable=None, total=None):
        cfg = getattr(self, "_progress_bar_config", {"disable": True})
        if iterable is not None:
            return tqdm(iterable, **cfg)
        return tqdm(total=total, **cfg)

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs
