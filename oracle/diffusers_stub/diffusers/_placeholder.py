"""Placeholder factory for diffusers symbols the hot path imports but never instantiates."""
import torch.nn as nn


def make_placeholder(name):
    def _init(self, *a, **k):
        raise NotImplementedError(
            f"diffusers stub: `{name}` is a placeholder (branch not reachable with the SD-1.5 "
            "config + inference_v2.yaml); it must not be instantiated"
        )

    return type(name, (nn.Module,), {"__init__": _init})
