"""Seeded inputs shared by `oracle/make_golden.py` (which runs the reference) and the parity tests
(which run the oracle / the HIP path).  Everything here is derived from fixed seeds with
`torch.Generator` on CPU, so it is reproducible on the GPU box without the reference."""
import torch


def _g(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def unet_case(small, f=4, h=16, w=16):
    """Inputs of one CFG-doubled denoising call + the ReferenceNet call that feeds it."""
    from aniportrait_amd import configs as C

    D = (C.SD15_UNET_SMALL if small else C.SD15_UNET)["cross_attention_dim"]
    clip = torch.randn((1, 1, D), generator=_g(11))
    ehs = torch.cat([torch.zeros_like(clip), clip], dim=0)
    ref_lat = torch.randn((1, 4, h, w), generator=_g(12))
    lat = torch.randn((1, 4, f, h, w), generator=_g(13)).repeat(2, 1, 1, 1, 1)
    pose = torch.randn((1, 3, f, h * 8, w * 8), generator=_g(14)).repeat(2, 1, 1, 1, 1)
    ref_pose = torch.randn((1, 3, h * 8, w * 8), generator=_g(15))
    return dict(ehs=ehs, ref_lat=ref_lat, lat=lat, pose=pose, ref_pose=ref_pose, t=959)


def vae_case(h=16, w=16):
    z = torch.randn((1, 4, h, w), generator=_g(21))
    x = torch.rand((1, 3, h * 8, w * 8), generator=_g(22)) * 2 - 1
    return dict(z=z, x=x)


PIPE_CASES = {
    # name: (H, W, L, steps, cfg, long, extra kwargs)
    "long_L4": (128, 128, 4, 3, 3.5, True, {}),
    "short_L4": (128, 128, 4, 2, 3.5, False, {}),
    "long_L10_ctx8": (128, 128, 10, 2, 3.5, True, dict(context_frames=8, context_overlap=2)),
    "long_L4_nocfg": (128, 128, 4, 2, 1.0, True, {}),
}


def pipe_inputs(name):
    from aniportrait_amd.synthetic import synth_latents, synth_pose_frames, synth_ref_image

    H, W, L, steps, cfg, long, kw = PIPE_CASES[name]
    return dict(H=H, W=W, L=L, steps=steps, cfg=cfg, long=long, kw=kw,
                poses=synth_pose_frames(L, H, W), ref_pose=synth_pose_frames(1, H, W, 999)[0],
                ref_image=synth_ref_image(H, W), latents=synth_latents(L, H // 8, W // 8, 42))


# Real-width (SD-1.5 / sd-vae-ft-mse widths) pipeline cases whose outputs the REFERENCE's own pipeline produced in the
# build container (oracle/make_golden_real_pipeline.py -> tests/golden/real_pipeline_<name>.pt); the GPU tests only
# compare against the committed fixture (no CPU oracle run on the GPU box).
REAL_PIPE_CASES = {
    # name: (H, W, L, steps, cfg, input seed, frames whose decoded pixels are stored)
    "c2_4step": (512, 512, 16, 4, 3.5, 2, tuple(range(16))),      # BASELINE configs[1] geometry, 4 of 25 DDIM steps
    "c5_1step": (768, 768, 16, 1, 3.5, 3, (0, 5, 10, 15)),        # BASELINE configs[4] geometry (96x96 latents)
    "l40_windows": (128, 128, 40, 3, 3.5, 4, (0, 11, 12, 23, 36, 39)),  # 4 windows / step incl. the wrap-around one
    "c2_25step": (512, 512, 16, 25, 3.5, 2, tuple(range(16))),    # BASELINE configs[1] IN FULL: the schedule bench.py times
    "c5_4step": (768, 768, 16, 4, 3.5, 3, (0, 5, 10, 15)),        # BASELINE configs[4] geometry, 4 of 25 DDIM steps
    # BASELINE configs[3] at its OWN geometry: 512x512, L=150 -> 13 overlapping 16-frame windows per step (the last one
    # wrapping around the clip end), 1 DDIM step; stored frames sit in plain, overlapping and wrap-around windows
    "c4_1step": (512, 512, 150, 1, 3.5, 5, (0, 5, 12, 75, 146, 149)),
    "c5_25step": (768, 768, 16, 25, 3.5, 3, (0, 5, 10, 15)),      # BASELINE configs[4] geometry with its WHOLE 25-step schedule
    # BASELINE configs[3] geometry over a MULTI-step schedule (round 6): the wrap-around window merge feeding the next step
    "c4_4step": (512, 512, 150, 4, 3.5, 5, (0, 5, 12, 75, 146, 149)),
}


def real_pipe_inputs(name):
    from aniportrait_amd.synthetic import synth_pose_frames, synth_ref_image

    H, W, L, steps, cfg, seed, frames = REAL_PIPE_CASES[name]
    return dict(H=H, W=W, L=L, steps=steps, cfg=cfg, seed=seed, frames=frames, gen_seed=42 + seed,
                ref_image=synth_ref_image(H, W, 1 + seed), poses=synth_pose_frames(L, H, W, 1234 + 100 * seed),
                ref_pose=synth_pose_frames(1, H, W, 999)[0])
