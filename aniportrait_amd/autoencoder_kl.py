"""`AutoencoderKL` (diffusers 0.24.0 API surface used by the reference: `scripts/pose2vid.py:59-61`,
`src/pipelines/pipeline_pose2vid_long.py:73,119-120,430`) over the HIP engine: sd-vae-ft-mse topology,
diffusers' state-dict key names, `.decode(z).sample`, `.encode(x).latent_dist.mean`.
"""
import torch

from . import engine, hipops as ops
from .modeling import BaseOutput, HipModel
from .params import vae_shapes


class DecoderOutput(BaseOutput):
    pass


class AutoencoderKLOutput(BaseOutput):
    pass


class _LatentMean:
    """The slice of DiagonalGaussianDistribution the pipeline reads (`latent_dist.mean`)."""

    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        raise NotImplementedError("sampling the VAE posterior is not on the pose2vid path (it uses .mean)")


class AutoencoderKL(HipModel):
    config_defaults = dict(
        in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",), up_block_types=("UpDecoderBlock2D",),
        block_out_channels=(64,), layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=32,
        sample_size=32, scaling_factor=0.18215, force_upcast=True,
    )

    @classmethod
    def _shapes(cls, cfg):
        if cfg["norm_num_groups"] != 32 or cfg["act_fn"] != "silu":
            raise NotImplementedError("AutoencoderKL: only norm_num_groups=32 / silu (sd-vae-ft-mse) is built")
        if any(t != "DownEncoderBlock2D" for t in cfg["down_block_types"]) or \
                any(t != "UpDecoderBlock2D" for t in cfg["up_block_types"]):
            raise NotImplementedError("AutoencoderKL: only DownEncoderBlock2D / UpDecoderBlock2D")
        return vae_shapes(cfg)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.use_slicing = False

    @classmethod
    def from_module(cls, vae):
        """Adopt a foreign (diffusers) AutoencoderKL: same config + state-dict key names."""
        if isinstance(vae, cls):
            return vae
        cfg = {k: v for k, v in dict(vae.config).items() if k in cls.config_defaults}
        m = cls(**cfg)
        m.load_state_dict(m._convert_legacy_keys(vae.state_dict()), strict=True)
        p = next(vae.parameters())
        return m.to(device=p.device, dtype=p.dtype)

    def _convert_legacy_keys(self, sd):
        """older checkpoints name the mid-block attention query/key/value/proj_attn"""
        ren = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
        out = {}
        for k, v in sd.items():
            if ".attentions." in k:
                for a, b in ren.items():
                    k = k.replace(a, b)
            out[k] = v
        return out

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    # channels-last entry points used by the pipeline
    def decode_nhwc(self, z, tap=None):
        return engine.vae_decode(self.packed(), self.config, z, tap)

    def encode_mean_nhwc(self, x):
        return engine.vae_encode_mean(self.packed(), self.config, x)

    def decode(self, z, return_dict=True, generator=None):
        """z (N, 4, h, w) -> sample (N, 3, 8h, 8w)"""
        N = z.shape[0]
        x = self.decode_nhwc(ops.ncfhw_to_nhwc(z.unsqueeze(2)))
        out = ops.nhwc_to_ncfhw(x, N, out_f32=(z.dtype == torch.float32)).squeeze(2)
        if not return_dict:
            return (out,)
        return DecoderOutput(sample=out)

    def encode(self, x, return_dict=True):
        N = x.shape[0]
        m = self.encode_mean_nhwc(ops.ncfhw_to_nhwc(x.unsqueeze(2)))
        mean = ops.nhwc_to_ncfhw(m, N, out_f32=(x.dtype == torch.float32)).squeeze(2)
        dist = _LatentMean(mean)
        if not return_dict:
            return (dist,)
        return AutoencoderKLOutput(latent_dist=dist)

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None):
        if sample_posterior:
            raise NotImplementedError("sample_posterior")
        return self.decode(self.encode(sample).latent_dist.mean, return_dict=return_dict)
