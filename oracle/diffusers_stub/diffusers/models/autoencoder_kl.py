"""Test-only stub of `diffusers.AutoencoderKL` (0.24.0 restated; sd-vae-ft-mse topology):
Encoder/Decoder built from `ResnetBlock2D`, one single-head mid-block `Attention` with an input
GroupNorm and residual connection, `Downsample2D(padding=0)` / `Upsample2D`.  Used by the
reference at `src/pipelines/pipeline_pose2vid_long.py:113-126,424-431`."""
from dataclasses import dataclass

import torch
import torch.nn as nn

from ..configuration_utils import ConfigMixin, register_to_config
from ..utils import BaseOutput
from .attention_processor import Attention
from .modeling_utils import ModelMixin
from .resnet import Downsample2D, ResnetBlock2D, Upsample2D


class UNetMidBlock2D(nn.Module):
    def __init__(self, in_channels, temb_channels, resnet_eps=1e-6, resnet_act_fn="swish", resnet_groups=32,
                 attention_head_dim=1, output_scale_factor=1.0, add_attention=True):
        super().__init__()
        def res():
            return ResnetBlock2D(in_channels=in_channels, out_channels=in_channels, temb_channels=temb_channels,
                                 eps=resnet_eps, groups=resnet_groups, non_linearity=resnet_act_fn,
                                 output_scale_factor=output_scale_factor)
        self.resnets = nn.ModuleList([res(), res()])
        self.attentions = nn.ModuleList([
            Attention(in_channels, heads=in_channels // attention_head_dim, dim_head=attention_head_dim,
                      rescale_output_factor=output_scale_factor, eps=resnet_eps, norm_num_groups=resnet_groups,
                      residual_connection=True, bias=True, upcast_softmax=True,
                      _from_deprecated_attn_block=True)])

    def forward(self, hidden_states, temb=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, temb=temb)
            hidden_states = resnet(hidden_states, temb)
        return hidden_states


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_downsample, resnet_eps, resnet_act_fn,
                 resnet_groups):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                          temb_channels=None, eps=resnet_eps, groups=resnet_groups, non_linearity=resnet_act_fn)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=0, name="op")]) if add_downsample else None

    def forward(self, hidden_states):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb=None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
        return hidden_states


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_upsample, resnet_eps, resnet_act_fn,
                 resnet_groups):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                          temb_channels=None, eps=resnet_eps, groups=resnet_groups, non_linearity=resnet_act_fn)
            for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) \
            if add_upsample else None

    def forward(self, hidden_states, temb=None):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb=temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups, act_fn,
                 double_z=True):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], kernel_size=3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList([])
        output_channel = block_out_channels[0]
        for i in range(len(block_out_channels)):
            input_channel = output_channel
            output_channel = block_out_channels[i]
            is_final = i == len(block_out_channels) - 1
            self.down_blocks.append(DownEncoderBlock2D(input_channel, output_channel, layers_per_block,
                                                       not is_final, 1e-6, act_fn, norm_num_groups))
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], None, resnet_eps=1e-6, resnet_act_fn=act_fn,
                                        resnet_groups=norm_num_groups, attention_head_dim=block_out_channels[-1])
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[-1], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[-1], 2 * out_channels if double_z else out_channels, 3,
                                  padding=1)

    def forward(self, sample):
        sample = self.conv_in(sample)
        for b in self.down_blocks:
            sample = b(sample)
        sample = self.mid_block(sample)
        return self.conv_out(self.conv_act(self.conv_norm_out(sample)))


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups, act_fn):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], kernel_size=3, stride=1, padding=1)
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], None, resnet_eps=1e-6, resnet_act_fn=act_fn,
                                        resnet_groups=norm_num_groups, attention_head_dim=block_out_channels[-1])
        self.up_blocks = nn.ModuleList([])
        rev = list(reversed(block_out_channels))
        output_channel = rev[0]
        for i in range(len(rev)):
            prev = output_channel
            output_channel = rev[i]
            is_final = i == len(rev) - 1
            self.up_blocks.append(UpDecoderBlock2D(prev, output_channel, layers_per_block + 1, not is_final,
                                                   1e-6, act_fn, norm_num_groups))
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    def forward(self, sample, latent_embeds=None):
        sample = self.conv_in(sample)
        sample = self.mid_block(sample, latent_embeds)
        for b in self.up_blocks:
            sample = b(sample, latent_embeds)
        return self.conv_out(self.conv_act(self.conv_norm_out(sample)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean


@dataclass
class AutoencoderKLOutput(BaseOutput):
    latent_dist: DiagonalGaussianDistribution = None


@dataclass
class DecoderOutput(BaseOutput):
    sample: torch.FloatTensor = None


class AutoencoderKL(ModelMixin, ConfigMixin):
    @register_to_config
    def __init__(self, in_channels=3, out_channels=3,
                 down_block_types=("DownEncoderBlock2D",), up_block_types=("UpDecoderBlock2D",),
                 block_out_channels=(64,), layers_per_block=1, act_fn="silu", latent_channels=4,
                 norm_num_groups=32, sample_size=32, scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block,
                               norm_num_groups, act_fn, double_z=True)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block,
                               norm_num_groups, act_fn)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.use_slicing = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def encode(self, x, return_dict=True):
        moments = self.quant_conv(self.encoder(x))
        return AutoencoderKLOutput(latent_dist=DiagonalGaussianDistribution(moments))

    def decode(self, z, return_dict=True, generator=None):
        dec = self.decoder(self.post_quant_conv(z))
        return DecoderOutput(sample=dec)
