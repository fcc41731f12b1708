"""Test-only stub of `diffusers.models.attention_processor` (0.24.0 behaviour restated from the
published source): `Attention` with `to_q/to_k/to_v` (bias optional), `to_out = [Linear, Dropout]`,
`scale = dim_head**-0.5`, optional GroupNorm on the input (VAE mid-block), and the two processors
the hot path reaches — `AttnProcessor2_0` (SDPA, the default) and `AttnProcessor` (explicit
baddbmm/softmax).  Call sites in the reference: `src/models/attention.py:111,133,323,339`,
`src/models/motion_module.py:280,345,377-383`, `src/models/mutual_self_attention.py:159-203`.
"""
from typing import Any, Callable, Dict, List, Optional, Tuple, Union  # noqa: F401 (star-imported by the reference)

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._placeholder import make_placeholder
from .lora import LoRACompatibleLinear


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None,
                 cross_attention_norm_num_groups=32, added_kv_proj_dim=None, norm_num_groups=None,
                 spatial_norm_dim=None, out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5,
                 rescale_output_factor=1.0, residual_connection=False, _from_deprecated_attn_block=False,
                 processor=None):
        super().__init__()
        assert cross_attention_norm is None and added_kv_proj_dim is None and spatial_norm_dim is None
        self.inner_dim = dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.dropout = dropout
        self._from_deprecated_attn_block = _from_deprecated_attn_block
        self.scale_qk = scale_qk
        self.scale = dim_head**-0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = None
        self.only_cross_attention = only_cross_attention
        self.group_norm = (nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = LoRACompatibleLinear(query_dim, self.inner_dim, bias=bias)
        self.to_k = LoRACompatibleLinear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = LoRACompatibleLinear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(self.inner_dim, query_dim, bias=out_bias),
                                     nn.Dropout(dropout)])
        if processor is None:
            processor = AttnProcessor2_0() if hasattr(F, "scaled_dot_product_attention") and scale_qk \
                else AttnProcessor()
        self.set_processor(processor)

    def set_processor(self, processor, _remove_lora=False):
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def set_use_memory_efficient_attention_xformers(self, use, attention_op=None):
        if use:
            raise ModuleNotFoundError("xformers is not available in the stub")

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    def batch_to_head_dim(self, tensor):
        head_size = self.heads
        batch_size, seq_len, dim = tensor.shape
        tensor = tensor.reshape(batch_size // head_size, head_size, seq_len, dim)
        return tensor.permute(0, 2, 1, 3).reshape(batch_size // head_size, seq_len, dim * head_size)

    def head_to_batch_dim(self, tensor, out_dim=3):
        head_size = self.heads
        batch_size, seq_len, dim = tensor.shape
        tensor = tensor.reshape(batch_size, seq_len, head_size, dim // head_size).permute(0, 2, 1, 3)
        if out_dim == 3:
            tensor = tensor.reshape(batch_size * head_size, seq_len, dim // head_size)
        return tensor

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        if attention_mask is None:
            baddbmm_input = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype,
                                        device=query.device)
            beta = 0
        else:
            baddbmm_input, beta = attention_mask, 1
        scores = torch.baddbmm(baddbmm_input, query, key.transpose(-1, -2), beta=beta, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        probs = scores.softmax(dim=-1)
        return probs.to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is None:
            return attention_mask
        raise NotImplementedError("attention masks are never passed on the hot path")


def _pre(attn, hidden_states, temb):
    residual = hidden_states
    input_ndim = hidden_states.ndim
    shape4 = None
    if input_ndim == 4:
        shape4 = hidden_states.shape
        b, c, h, w = shape4
        hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
    if attn.group_norm is not None:
        hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
    return residual, input_ndim, shape4, hidden_states


def _post(attn, hidden_states, residual, input_ndim, shape4):
    hidden_states = attn.to_out[0](hidden_states)
    hidden_states = attn.to_out[1](hidden_states)
    if input_ndim == 4:
        b, c, h, w = shape4
        hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, h, w)
    if attn.residual_connection:
        hidden_states = hidden_states + residual
    return hidden_states / attn.rescale_output_factor


class AttnProcessor:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale=1.0):
        residual, input_ndim, shape4, hidden_states = _pre(attn, hidden_states, temb)
        assert attention_mask is None
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        query = attn.head_to_batch_dim(query)
        key = attn.head_to_batch_dim(key)
        value = attn.head_to_batch_dim(value)
        probs = attn.get_attention_scores(query, key, None)
        hidden_states = torch.bmm(probs, value)
        hidden_states = attn.batch_to_head_dim(hidden_states)
        return _post(attn, hidden_states, residual, input_ndim, shape4)


class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale=1.0):
        residual, input_ndim, shape4, hidden_states = _pre(attn, hidden_states, temb)
        assert attention_mask is None
        batch_size = hidden_states.shape[0]
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        inner_dim = key.shape[-1]
        head_dim = inner_dim // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0,
                                                       is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = hidden_states.to(query.dtype)
        return _post(attn, hidden_states, residual, input_ndim, shape4)


AttnAddedKVProcessor = make_placeholder("AttnAddedKVProcessor")
AttnAddedKVProcessor2_0 = make_placeholder("AttnAddedKVProcessor2_0")
XFormersAttnProcessor = make_placeholder("XFormersAttnProcessor")
ADDED_KV_ATTENTION_PROCESSORS = (AttnAddedKVProcessor, AttnAddedKVProcessor2_0)
CROSS_ATTENTION_PROCESSORS = (AttnProcessor, AttnProcessor2_0)
AttentionProcessor = Union[AttnProcessor, AttnProcessor2_0]
