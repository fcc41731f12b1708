"""Test-only stub of the `diffusers.utils` names the reference imports."""
import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch
from packaging import version as _version

WEIGHTS_NAME = "diffusion_pytorch_model.bin"
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
USE_PEFT_BACKEND = False


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _pylogging.getLogger(name)


logging = _Logging()


class BaseOutput(OrderedDict):
    """Dataclass-backed ordered dict; supports attribute, key and integer access."""

    def __post_init__(self):
        if is_dataclass(self):
            for f in fields(self):
                v = getattr(self, f.name)
                if v is not None:
                    OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(OrderedDict.__getitem__(self, k) for k in self.keys())


def deprecate(*args, **kwargs):
    return None


def is_torch_version(op, ver):
    cur = _version.parse(torch.__version__.split("+")[0])
    ref = _version.parse(ver)
    return {">=": cur >= ref, ">": cur > ref, "<=": cur <= ref, "<": cur < ref, "==": cur == ref}[op]


def is_accelerate_available():
    return False


def scale_lora_layers(model, weight):
    return None


def unscale_lora_layers(model, weight=None):
    return None
