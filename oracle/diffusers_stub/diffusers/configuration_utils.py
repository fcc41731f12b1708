"""Minimal restatement of diffusers 0.24.0 `configuration_utils` (test-only stub).

Covers what the reference uses: `@register_to_config` capturing ctor kwargs (with signature
defaults) into `self.config`, attribute fall-through (`model.in_channels`,
`src/pipelines/pipeline_pose2vid_long.py:408`), `load_config` (json) and `from_config`
(`src/models/unet_3d.py:603-619`).
"""
import functools
import inspect
import json
import os
from collections import OrderedDict


class FrozenDict(OrderedDict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in self.items():
            object.__setattr__(self, k, v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


class ConfigMixin:
    config_name = "config.json"
    ignore_for_config = []

    def register_to_config(self, **kwargs):
        kwargs.pop("kwargs", None)
        prev = dict(getattr(self, "_internal_dict", {}))
        prev.update(kwargs)
        object.__setattr__(self, "_internal_dict", FrozenDict(prev))

    @property
    def config(self):
        return self._internal_dict

    def __getattr__(self, name):
        d = self.__dict__.get("_internal_dict", None)
        if d is not None and name in d and name not in self.__dict__:
            return d[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    @classmethod
    def load_config(cls, path, subfolder=None, **kwargs):
        path = str(path)
        if os.path.isdir(path):
            if subfolder is not None:
                path = os.path.join(path, subfolder)
            path = os.path.join(path, cls.config_name)
        with open(path, "r") as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **kwargs):
        params = inspect.signature(cls.__init__).parameters
        init = {k: v for k, v in dict(config).items() if k in params and not k.startswith("_")}
        for k in list(kwargs):
            if k in params:
                init[k] = kwargs.pop(k)
        return cls(**init)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        names = [n for n in sig.parameters if n != "self"]
        cfg = {n: p.default for n, p in sig.parameters.items()
               if n != "self" and p.default is not inspect.Parameter.empty}
        for n, a in zip(names, args):
            cfg[n] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        init(self, *args, **{k: v for k, v in kwargs.items() if not k.startswith("_")})
        cfg = {k: v for k, v in cfg.items() if k not in getattr(self, "ignore_for_config", [])}
        self.register_to_config(**cfg)

    return inner
