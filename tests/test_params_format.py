"""State-dict format contract: names/shapes derived from configs (aniportrait_amd/params.py) against
manifests dumped from the REFERENCE's own modules (tests/golden/shapes_*.json, made by
oracle/make_golden.py), and the host classes built on them."""
import pytest
import torch

from util import load_manifest


@pytest.mark.parametrize("small", [True, False])
def test_shapes_match_reference_manifests(small):
    from aniportrait_amd import configs as C
    from aniportrait_amd import params as P
    man = load_manifest(small)
    ch0 = (C.SD15_UNET_SMALL if small else C.SD15_UNET)["block_out_channels"][0]
    cases = {
        "denoising_unet": P.unet_shapes(C.unet3d_kwargs(small), True),
        "reference_unet": P.unet_shapes(C.unet2d_kwargs(small), False),
        "vae": P.vae_shapes(C.SD_VAE_SMALL if small else C.SD_VAE_FT_MSE),
        "pose_guider": P.pose_guider_shapes(ch0, True),
    }
    for k, (params, bufs) in cases.items():
        assert {n: tuple(s) for n, s in params.items()} == {n: tuple(s) for n, s in man[k]["params"].items()}, k
        assert {n: tuple(s) for n, s in bufs.items()} == {n: tuple(s) for n, s in man[k]["buffers"].items()}, k
        assert list(params) and len(set(params)) == len(params)


def test_reference_attention_pairing_order():
    """ReferenceAttentionControl sorts DFS-ordered blocks (down, up, mid) by -width, stably
    (src/models/mutual_self_attention.py:321-337)."""
    from aniportrait_amd import configs as C
    from aniportrait_amd import params as P
    p3 = P.transformer_block_paths(P.unet_shapes(C.unet3d_kwargs(False), True)[0])
    p2 = P.transformer_block_paths(P.unet_shapes(C.unet2d_kwargs(False), False)[0])
    assert p3 == p2 and len(p3) == 16
    assert p3[0] == "down_blocks.2.attentions.0.transformer_blocks.0"
    assert p3[5] == "mid_block.attentions.0.transformer_blocks.0"       # 1280-wide: down2 x2, up1 x3, mid
    assert p3[-1] == "up_blocks.3.attentions.2.transformer_blocks.0"


def test_modules_accept_reference_state_dicts():
    from aniportrait_amd import configs as C
    from aniportrait_amd.autoencoder_kl import AutoencoderKL
    from aniportrait_amd.pose_guider import PoseGuider
    from aniportrait_amd.unet import UNet2DConditionModel, UNet3DConditionModel
    man = load_manifest(True)
    u3 = UNet3DConditionModel(**C.unet3d_kwargs(True))
    u2 = UNet2DConditionModel(**C.unet2d_kwargs(True))
    vae = AutoencoderKL(**C.SD_VAE_SMALL)
    pg = PoseGuider(noise_latent_channels=64)
    for m, k in ((u3, "denoising_unet"), (u2, "reference_unet"), (vae, "vae"), (pg, "pose_guider")):
        want = set(man[k]["params"]) | set(man[k]["buffers"])
        assert set(m.state_dict()) == want, k
        sd = {n: torch.zeros(s) for n, s in man[k]["params"].items()}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and set(missing) <= set(man[k]["buffers"])
    # the ReferenceNet has no conv_out (reference_unet.pth is loaded strict, scripts/pose2vid.py:95-97)
    assert not any(n.startswith("conv_out") or n.startswith("conv_norm_out") for n in u2.state_dict())
    assert u3.in_channels == 4 and u3.config.block_out_channels == (64, 128, 256, 256)
    assert u3.dtype == torch.float32 and u3.device.type == "cpu"
    with pytest.raises(RuntimeError):
        u2.load_state_dict({"conv_out.weight": torch.zeros(4, 64, 3, 3)}, strict=True)


def test_from_config_filters_and_rejects_unbuilt_variants():
    from aniportrait_amd import configs as C
    from aniportrait_amd.unet import UNet3DConditionModel
    cfg = dict(C.SD15_UNET_SMALL, _class_name="UNet2DConditionModel", _diffusers_version="0.6.0", some_new_field=1)
    m = UNet3DConditionModel.from_config(cfg, **C.INFERENCE_V2)
    assert m.config.use_motion_module and m.config.motion_module_kwargs["temporal_position_encoding_max_len"] == 32
    # inference_v1.yaml (GroupNorm statistics across frames) is built since round 3
    v1 = UNet3DConditionModel(**dict(C.unet3d_kwargs(True), use_inflated_groupnorm=False, motion_module_mid_block=False))
    assert v1.config.use_inflated_groupnorm is False and not any(k.startswith("mid_block.motion") for k in v1.state_dict())
    with pytest.raises(NotImplementedError):
        UNet3DConditionModel(**dict(C.unet3d_kwargs(True), use_linear_projection=True))


def test_vae_legacy_attention_keys():
    from aniportrait_amd import configs as C
    from aniportrait_amd.autoencoder_kl import AutoencoderKL
    vae = AutoencoderKL(**C.SD_VAE_SMALL)
    sd = vae.state_dict()
    ren = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}
    old = {}
    for k, v in sd.items():
        if ".attentions." in k:
            for a, b in ren.items():
                k = k.replace(a, b)
        old[k] = v
    assert set(old) != set(sd)
    assert set(vae._convert_legacy_keys(old)) == set(sd)
