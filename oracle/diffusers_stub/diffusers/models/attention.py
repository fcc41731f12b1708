"""Test-only stub of `diffusers.models.attention` (0.24.0 restated): `FeedForward` with the
GEGLU activation (exact erf GELU), `AdaLayerNorm` constructible-but-unused."""
from typing import Any, Callable, Dict, Optional  # noqa: F401

import torch  # noqa: F401
import torch.nn as nn
import torch.nn.functional as F

from .._placeholder import make_placeholder
from .attention_processor import Attention  # noqa: F401  (re-export, reference imports it from here)
from .lora import LoRACompatibleLinear


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2)

    def gelu(self, gate):
        return F.gelu(gate)

    def forward(self, hidden_states, scale=1.0):
        hidden_states, gate = self.proj(hidden_states, scale).chunk(2, dim=-1)
        return hidden_states * self.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        if activation_fn != "geglu":
            raise NotImplementedError("only geglu is reachable on the hot path")
        self.net = nn.ModuleList([GEGLU(dim, inner_dim), nn.Dropout(dropout),
                                  LoRACompatibleLinear(inner_dim, dim_out)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states, scale=1.0):
        for module in self.net:
            if isinstance(module, (LoRACompatibleLinear, GEGLU)):
                hidden_states = module(hidden_states, scale)
            else:
                hidden_states = module(hidden_states)
        return hidden_states


AdaLayerNorm = make_placeholder("AdaLayerNorm")
AdaLayerNormZero = make_placeholder("AdaLayerNormZero")
GatedSelfAttentionDense = make_placeholder("GatedSelfAttentionDense")
