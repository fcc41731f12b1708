"""Literal configurations of the networks on the pose2vid hot path.

* ``SD15_UNET``      — the Stable-Diffusion-1.5 ``unet/config.json`` fields the reference reads
                        (reference consumer: ``src/models/unet_3d.py:36-83``).
* ``INFERENCE_V2``   — ``unet_additional_kwargs`` of ``configs/inference/inference_v2.yaml:1-22``.
* ``DDIM_V2``        — ``noise_scheduler_kwargs`` of ``configs/inference/inference_v2.yaml:24-33``.
* ``DDIM_V1`` / ``unet3d_kwargs_v1`` — ``configs/inference/inference_v1.yaml``: epsilon prediction, leading spacing, no
                        zero-SNR rescale; nn.GroupNorm over the sample's frames, no mid-block motion module, 24-frame pe table.
* ``SD_VAE_FT_MSE``  — sd-vae-ft-mse ``config.json``.
* ``*_SMALL``        — reduced-width variants with the same topology, used by CPU tests and small
                        parity cases (8 heads kept, head dims 8/16/32 instead of 40/80/160).
"""
import copy

SD15_UNET = dict(
    sample_size=64,
    in_channels=4,
    out_channels=4,
    center_input_sample=False,
    flip_sin_to_cos=True,
    freq_shift=0,
    block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2,
    downsample_padding=1,
    mid_block_scale_factor=1,
    act_fn="silu",
    norm_num_groups=32,
    norm_eps=1e-5,
    cross_attention_dim=768,
    attention_head_dim=8,
    use_linear_projection=False,
)

INFERENCE_V2 = dict(
    use_inflated_groupnorm=True,
    unet_use_cross_frame_attention=False,
    unet_use_temporal_attention=False,
    use_motion_module=True,
    motion_module_resolutions=(1, 2, 4, 8),
    motion_module_mid_block=True,
    motion_module_decoder_only=False,
    motion_module_type="Vanilla",
    motion_module_kwargs=dict(
        num_attention_heads=8,
        num_transformer_block=1,
        attention_block_types=("Temporal_Self", "Temporal_Self"),
        temporal_position_encoding=True,
        temporal_position_encoding_max_len=32,
        temporal_attention_dim_div=1,
    ),
)

DDIM_V2 = dict(
    beta_start=0.00085,
    beta_end=0.012,
    beta_schedule="linear",
    clip_sample=False,
    steps_offset=1,
    prediction_type="v_prediction",
    rescale_betas_zero_snr=True,
    timestep_spacing="trailing",
)

DDIM_V1 = dict(          # configs/inference/inference_v1.yaml:18-23 (everything else at DDIMScheduler's defaults)
    beta_start=0.00085,
    beta_end=0.012,
    beta_schedule="linear",
    steps_offset=1,
    clip_sample=False,
)

SD_VAE_FT_MSE = dict(
    in_channels=3,
    out_channels=3,
    down_block_types=("DownEncoderBlock2D",) * 4,
    up_block_types=("UpDecoderBlock2D",) * 4,
    block_out_channels=(128, 256, 512, 512),
    layers_per_block=2,
    act_fn="silu",
    latent_channels=4,
    norm_num_groups=32,
    sample_size=256,
    scaling_factor=0.18215,
)


def _with(base, **kw):
    d = copy.deepcopy(base)
    d.update(kw)
    return d


SD15_UNET_SMALL = _with(SD15_UNET, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)
SD_VAE_SMALL = _with(SD_VAE_FT_MSE, block_out_channels=(64, 128, 256, 256))

# CLIP image encoder: lambdalabs sd-image-variations `image_encoder` = ViT-L/14, projection 768.
CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=224, patch_size=14, projection_dim=768)
CLIP_SMALL = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                  image_size=224, patch_size=32, projection_dim=64)


def unet3d_kwargs(small=False):
    """ctor kwargs of ``UNet3DConditionModel`` (SD-1.5 config + inference_v2 additions)."""
    d = copy.deepcopy(SD15_UNET_SMALL if small else SD15_UNET)
    d.update(copy.deepcopy(INFERENCE_V2))
    return d


def unet3d_kwargs_v1(small=False):
    """ctor kwargs of ``UNet3DConditionModel`` under configs/inference/inference_v1.yaml:1-16"""
    d = unet3d_kwargs(small)
    d.update(use_inflated_groupnorm=False, motion_module_mid_block=False)
    d["motion_module_kwargs"]["temporal_position_encoding_max_len"] = 24
    return d


def unet2d_kwargs(small=False):
    return copy.deepcopy(SD15_UNET_SMALL if small else SD15_UNET)
