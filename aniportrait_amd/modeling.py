"""Host-side base classes: the slice of diffusers' ModelMixin / ConfigMixin / BaseOutput surface the
reference's scripts and pipelines touch (`scripts/pose2vid.py:59-110`, `src/models/unet_3d.py:582-673`),
implemented without diffusers so the package imports on a bare ROCm image.

A `HipModel` is an `nn.Module` whose children are parameter containers spelling the reference's
state-dict names (`params.build_tree`); arithmetic is delegated to `engine` on packed device copies
of the weights (`engine.PackedNet`), rebuilt lazily whenever the parameters change.
"""
import inspect
import json
import os
from collections import OrderedDict

import torch
from torch import nn

from . import _lib
from .engine import PackedNet

WEIGHTS_NAME = "diffusion_pytorch_model.bin"
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
CONFIG_NAME = "config.json"


class FrozenConfig(OrderedDict):
    """dict with attribute access (diffusers FrozenDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        raise AttributeError("config is frozen")


class BaseOutput(OrderedDict):
    """diffusers BaseOutput: attribute + index access."""

    def __init__(self, **kw):
        super().__init__(**kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            return list(self.values())[k]
        return super().__getitem__(k)

    def to_tuple(self):
        return tuple(self.values())


def load_state_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd["state_dict"] if isinstance(sd, dict) and "state_dict" in sd else sd


class HipModel(nn.Module):
    """Base of the HIP-backed networks.  Subclasses define `config_defaults` and `_shapes(config)`."""

    config_name = CONFIG_NAME
    config_defaults = {}
    ignore_for_config = ()

    def __init__(self, **kwargs):
        super().__init__()
        unknown = set(kwargs) - set(self.config_defaults)
        cfg = dict(self.config_defaults)
        cfg.update({k: v for k, v in kwargs.items() if k in self.config_defaults})
        object.__setattr__(self, "_config", FrozenConfig(cfg))
        object.__setattr__(self, "_unused_config", {k: kwargs[k] for k in unknown})
        object.__setattr__(self, "_packed", None)
        params, buffers = self._shapes(cfg)
        from .params import build_tree
        build_tree(self, params, buffers)

    # -- config ------------------------------------------------------------------------------------
    @property
    def config(self):
        return self._config

    def __getattr__(self, name):
        # diffusers lets `model.in_channels` fall through to the config (used at
        # src/pipelines/pipeline_pose2vid_long.py:408)
        try:
            return super().__getattr__(name)
        except AttributeError:
            cfg = self.__dict__.get("_config")
            if cfg is not None and name in cfg:
                return cfg[name]
            raise

    @classmethod
    def load_config(cls, path, subfolder=None, **kw):
        if subfolder is not None:
            path = os.path.join(path, subfolder)
        f = path if os.path.isfile(path) else os.path.join(path, cls.config_name)
        if not os.path.isfile(f):
            raise RuntimeError(f"{f} does not exist")
        with open(f) as fh:
            return json.load(fh)

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kwargs)
        accepted = set(cls.config_defaults)
        return cls(**{k: v for k, v in cfg.items() if k in accepted})

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, torch_dtype=None, **kwargs):
        """config.json + diffusion_pytorch_model.{safetensors,bin}; unexpected keys are tolerated (the
        ReferenceNet drops SD's conv_out), then `.eval()` as diffusers does."""
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        model = cls.from_config(cls.load_config(path), **kwargs)
        for fn in (SAFETENSORS_WEIGHTS_NAME, WEIGHTS_NAME):
            f = os.path.join(path, fn)
            if os.path.isfile(f):
                sd = load_state_file(f)
                break
        else:
            raise RuntimeError(f"no weights found in {path} ({SAFETENSORS_WEIGHTS_NAME} / {WEIGHTS_NAME})")
        sd = model._convert_legacy_keys(sd)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing:
            raise RuntimeError(f"{cls.__name__}.from_pretrained: missing keys, e.g. {missing[:5]}")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def _convert_legacy_keys(self, sd):
        return sd

    def save_config(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump({"_class_name": type(self).__name__, **{k: v for k, v in self.config.items()}}, f, indent=2,
                      default=list)

    # -- dtype / device ----------------------------------------------------------------------------
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    # -- packed weights ----------------------------------------------------------------------------
    def _invalidate(self):
        object.__setattr__(self, "_packed", None)

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        self._invalidate()
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def packed(self):
        """Kernel-ready weights on this module's device (GPU only: there is no CPU path)."""
        dev = self.device
        if dev.type != "cuda":
            raise _lib.HipLibraryError(
                f"{type(self).__name__} is on {dev}: the pose2vid hot path only runs on an MI355X through "
                "libaniportrait_hip.so (no CPU / PyTorch fallback); move the module with .to('cuda')")
        _lib.load()
        if self._packed is None or self._packed.device != dev:
            object.__setattr__(self, "_packed", PackedNet(self.state_dict(), dev))
        return self._packed


def ctor_kwargs(cls):
    return [p for p in inspect.signature(cls.__init__).parameters if p != "self"]
