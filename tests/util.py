"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
MODEL_KEYS = ["denoising_unet", "reference_unet", "vae", "pose_guider"]


def load_manifest(small):
    with open(os.path.join(GOLD, "shapes_small.json" if small else "shapes_real.json")) as f:
        return json.load(f)


def oracle_state_dicts(small, keys=MODEL_KEYS, seed=0):
    """name-hash synthetic fp32 state-dicts with the reference's key names (params only; the
    oracle recomputes the positional-encoding buffers)."""
    from aniportrait_amd.synthetic import synth_state_dict

    man = load_manifest(small)
    return {k: synth_state_dict(man[k]["params"], seed, prefix=k + ".") for k in keys}


def load_golden(name):
    return torch.load(os.path.join(GOLD, name), map_location="cpu")


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def build_hip_models(small, keys=("denoising_unet", "reference_unet", "vae", "pose_guider"), dtype=None, device="cuda"):
    """The product's modules (aniportrait_amd) filled with the same name-hash synthetic weights as the
    oracle / the reference run that produced the goldens.  Returns (modules, oracle state-dicts)."""
    import torch as _t

    from aniportrait_amd import configs as C
    from aniportrait_amd.autoencoder_kl import AutoencoderKL
    from aniportrait_amd.pose_guider import PoseGuider
    from aniportrait_amd.unet import UNet2DConditionModel, UNet3DConditionModel

    dtype = dtype or _t.float16
    sds = oracle_state_dicts(small, keys=list(keys))
    m = {}
    if "denoising_unet" in keys:
        m["denoising_unet"] = UNet3DConditionModel(**C.unet3d_kwargs(small))
    if "reference_unet" in keys:
        m["reference_unet"] = UNet2DConditionModel(**C.unet2d_kwargs(small))
    if "vae" in keys:
        m["vae"] = AutoencoderKL(**(C.SD_VAE_SMALL if small else C.SD_VAE_FT_MSE))
    if "pose_guider" in keys:
        ch0 = (C.SD15_UNET_SMALL if small else C.SD15_UNET)["block_out_channels"][0]
        m["pose_guider"] = PoseGuider(noise_latent_channels=ch0, use_ca=True)
    for k in m:
        missing, unexpected = m[k].load_state_dict(sds[k], strict=False)
        assert not unexpected, unexpected[:3]
        assert all(x.endswith((".pe", "running_mean", "running_var", "num_batches_tracked")) for x in missing), missing[:3]
        m[k] = m[k].to(device, dtype)
    return m, sds


def small_clip_encoder(device="cuda"):
    """tiny CLIP vision tower (fp32) with name-hash weights, as in oracle/ref_harness.build_models"""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from aniportrait_amd import configs as C
    from aniportrait_amd.synthetic import fill_module_
    enc = CLIPVisionModelWithProjection(CLIPVisionConfig(**C.CLIP_SMALL))
    return fill_module_(enc, 0, "image_encoder.").eval().to(device)


def psnr(a, b, peak=1.0):
    import math
    mse = torch.mean((a.double().cpu() - b.double().cpu()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)


_MODEL_CACHE = {}


def build_hip_models_cached(small, device="cuda"):
    """`build_hip_models` once per test session and width (the real-width state-dicts take a minute to synthesise)"""
    key = (bool(small), device)
    if key not in _MODEL_CACHE:
        _MODEL_CACHE[key] = build_hip_models(small, device=device)
    return _MODEL_CACHE[key]


def clip_encoder_for(small, device="cuda"):
    """tiny CLIP vision tower (fp32, name-hash weights) projecting to the UNet's cross_attention_dim: the CLIP image
    encoder is outside the HIP scope and only its (1, D) output enters the path"""
    if small:
        return small_clip_encoder(device)
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from aniportrait_amd import configs as C
    from aniportrait_amd.synthetic import fill_module_
    enc = CLIPVisionModelWithProjection(CLIPVisionConfig(**dict(C.CLIP_SMALL, projection_dim=768)))
    return fill_module_(enc, 0, "image_encoder768.").eval().to(device)


def oracle_threads():
    """bound the CPU oracle's thread pool: hundreds of schedulable cores on the GPU box oversubscribe small GEMMs"""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(64, n)))
    return torch.get_num_threads()
