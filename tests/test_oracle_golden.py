"""The CPU oracle (oracle/ref_torch.py) against golden vectors produced by the REFERENCE's own
code (oracle/make_golden.py).  fp32 vs fp32: tolerance 2e-4 relative to the tensor's max (observed
~1e-5; the two sides only differ in op ordering)."""
import json
import os

import pytest
import torch

from util import GOLD, load_golden, oracle_state_dicts, rel_err

TOL = 2e-4


@pytest.fixture(scope="module")
def small():
    from aniportrait_amd import configs as C
    return dict(sds=oracle_state_dicts(True), ucfg=C.unet3d_kwargs(True), vcfg=C.SD_VAE_SMALL,
                gold=load_golden("small_models.pt"))


def test_context_windows_match_reference():
    from oracle import ref_torch as O
    with open(os.path.join(GOLD, "context_windows.json")) as f:
        win = json.load(f)
    for L in (4, 16, 17, 24, 46, 150):
        assert O.uniform_windows(0, L, 16, 1, 4) == win[str(L)]
    assert O.uniform_windows(0, 10, 8, 1, 2) == win["10_ctx8_ov2"]
    assert [len(win[str(L)]) for L in (4, 16, 17, 24, 46, 150)] == [1, 1, 2, 2, 4, 13]
    assert win["150"][-1] == list(range(144, 150)) + list(range(0, 10))


def test_refnet_banks(small):
    from golden_inputs import unet_case
    from oracle import ref_torch as O
    c = unet_case(True)
    with torch.no_grad():
        banks = O.refnet_forward(small["sds"]["reference_unet"], small["ucfg"], c["ref_lat"].repeat(2, 1, 1, 1),
                                 0, c["ehs"])
    gold = {k[5:]: v for k, v in small["gold"].items() if k.startswith("bank/")}
    assert set(gold) == set(banks) and len(banks) == 16
    for k in gold:
        # banks are fp16-rounded on both sides: allow one fp16 ulp of the max
        assert rel_err(banks[k], gold[k].float()) < 2e-3, k


def test_pose_guider(small):
    from golden_inputs import unet_case
    from oracle import ref_torch as O
    c = unet_case(True)
    with torch.no_grad():
        fea = O.pose_guider(small["sds"]["pose_guider"], c["pose"], c["ref_pose"])
    for i, f in enumerate(fea):
        assert rel_err(f, small["gold"][f"pose_fea/{i}"]) < TOL


@pytest.mark.parametrize("with_pose", [True, False])
def test_unet3d_reference_attention_cfg(small, with_pose):
    from golden_inputs import unet_case
    from oracle import ref_torch as O
    c = unet_case(True)
    g = small["gold"]
    banks = {k[5:]: v.float() for k, v in g.items() if k.startswith("bank/")}
    pose = [g[f"pose_fea/{i}"] for i in range(5)] if with_pose else None
    with torch.no_grad():
        out = O.unet3d_forward(small["sds"]["denoising_unet"], small["ucfg"], c["lat"], c["t"], c["ehs"], pose,
                               banks, True)
    assert rel_err(out, g["unet_out" if with_pose else "unet_out_nopose"]) < TOL


def test_vae(small):
    from golden_inputs import vae_case
    from oracle import ref_torch as O
    v = vae_case(16, 16)
    with torch.no_grad():
        dec = O.vae_decode(small["sds"]["vae"], small["vcfg"], v["z"])
        enc = O.vae_encode_mean(small["sds"]["vae"], small["vcfg"], v["x"])
    assert rel_err(dec, small["gold"]["vae_dec"]) < TOL
    assert rel_err(enc, small["gold"]["vae_enc"]) < TOL


@pytest.mark.parametrize("case", ["long_L4", "short_L4", "long_L10_ctx8", "long_L4_nocfg"])
def test_pipeline_matches_reference(small, case):
    """End-to-end: identical latents/seed -> decoded frames.  Bar: PSNR >= 60 dB vs the reference's
    fp32 frames stored as fp16 (fp16 storage alone limits PSNR to ~70 dB)."""
    from golden_inputs import pipe_inputs
    from oracle import ref_torch as O
    gold = load_golden("small_pipeline.pt")
    i = pipe_inputs(case)
    cfgs = {"unet": small["ucfg"], "vae": small["vcfg"]}
    vid = O.pose2vid(small["sds"], cfgs, gold[case + "/clip_embeds"], i["ref_image"], list(i["poses"]),
                     i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"], i["latents"], long=i["long"],
                     **i["kw"])
    ref = gold[case + "/video_f16"].float()
    assert vid.shape == ref.shape
    assert O.psnr(vid, ref) >= 60.0
    assert abs(vid.double().mean().item() - gold[case + "/video_mean"].item()) < 1e-4


def test_vae_real_width():
    """sd-vae-ft-mse widths (128,256,512,512), 32x32 latents -> 256x256 frame."""
    from aniportrait_amd import configs as C
    from golden_inputs import vae_case
    from oracle import ref_torch as O
    sds = oracle_state_dicts(False, keys=["vae"])
    gold = load_golden("real_models.pt")
    v = vae_case(32, 32)
    with torch.no_grad():
        dec = O.vae_decode(sds["vae"], C.SD_VAE_FT_MSE, v["z"])
        enc = O.vae_encode_mean(sds["vae"], C.SD_VAE_FT_MSE, v["x"])
    assert rel_err(dec, gold["vae_dec"]) < TOL
    assert rel_err(enc, gold["vae_enc"]) < TOL


def _v1_cfg():
    import copy

    from aniportrait_amd import configs as C
    kw = copy.deepcopy(C.unet3d_kwargs(True))
    kw.update(use_inflated_groupnorm=False, motion_module_mid_block=False)
    kw["motion_module_kwargs"]["temporal_position_encoding_max_len"] = 24
    return kw


@torch.no_grad()
def test_oracle_inference_v1_groupnorm_variant_matches_reference():
    """configs/inference/inference_v1.yaml: use_inflated_groupnorm absent -> ResnetBlock3D norm1 / norm2 and conv_norm_out are
    nn.GroupNorm over the 5-D tensor (statistics across the sample's frames: src/models/resnet.py:161-164,186-193,
    src/models/unet_3d.py:237-246), no mid-block motion module — golden made by the reference's own UNet3DConditionModel"""
    from golden_inputs import unet_case
    from oracle import ref_torch as O
    from util import load_golden, oracle_state_dicts, rel_err
    gold = load_golden("small_models_v1.pt")
    cfg = _v1_cfg()
    sd = oracle_state_dicts(True, keys=["denoising_unet"])["denoising_unet"]
    sd = {k: v for k, v in sd.items() if not k.startswith("mid_block.motion_modules")}
    c = unet_case(True)
    out = O.unet3d_forward(sd, cfg, c["lat"], c["t"], c["ehs"], None, None, True)
    assert rel_err(out, gold["unet_out_v1"]) < 2e-4
    assert rel_err(out, gold["unet_out_v1_if_inflated"]) > 1e-2      # the variant is live: per-frame norms differ
