"""TEST INFRASTRUCTURE — runs the *reference's own* Python files (AniPortrait @ 2024_08_07, read
in place from /root/reference) on PyTorch-CPU fp32 against the test-only diffusers stub in
`oracle/diffusers_stub/`.  Only usable where /root/reference exists (the build container); it is
how golden fixtures are produced (`oracle/make_golden.py`) and how `oracle/ref_torch.py` is
validated.  Never imported by the product.

Parity status: the reference files run unmodified, but `diffusers==0.24.0` underneath is a
restatement => "parity unpinned" at that boundary (see oracle/diffusers_stub/README.md).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ANIP_REFERENCE_ROOT", "/root/reference")
_STUB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusers_stub")
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models"))


def setup():
    """Make `import src.models...` resolve to the reference and `import diffusers` to the stub."""
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    # transformers must resolve its image backends BEFORE the empty torchvision stub appears
    from transformers import CLIPImageProcessor  # noqa: F401

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
    # the reference's checkout must come BEFORE this repository on sys.path: both hold a `src/` namespace package
    # (no __init__.py on either side), so path order decides which `src.models.*` / `src.pipelines.*` files win
    for p in (_STUB, REFERENCE_ROOT):
        while p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    if _REPO not in sys.path:
        sys.path.append(_REPO)
    # drop whatever `src*` modules were imported before (e.g. the repo's drop-in shim)
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        f = getattr(sys.modules[k], "__file__", None) or ""
        if not f.startswith(REFERENCE_ROOT):
            del sys.modules[k]
    import importlib
    importlib.invalidate_caches()
    assert_reference("src.pipelines.context")


def assert_reference(*module_names):
    """every named module must resolve to a file under REFERENCE_ROOT — guards the golden recipe against silently
    comparing the product with itself"""
    import importlib
    for name in module_names:
        m = importlib.import_module(name)
        f = getattr(m, "__file__", None) or ""
        if not os.path.abspath(f).startswith(os.path.abspath(REFERENCE_ROOT) + os.sep):
            raise RuntimeError(f"oracle/ref_harness: `{name}` resolved to {f!r}, not to the reference under "
                               f"{REFERENCE_ROOT} — refusing to produce 'reference' outputs from the product's own code")


def build_models(small=True, seed=0, with_clip=True, dtype=None):
    """Reference-class instances with name-hash synthetic weights (fp32, fp16-representable)."""
    import torch
    setup()
    from diffusers import AutoencoderKL, DDIMScheduler
    from src.models.pose_guider import PoseGuider
    from src.models.unet_2d_condition import UNet2DConditionModel
    from src.models.unet_3d import UNet3DConditionModel
    assert_reference("src.models.pose_guider", "src.models.unet_2d_condition", "src.models.unet_3d",
                     "src.models.mutual_self_attention")

    from aniportrait_amd import configs as C
    from aniportrait_amd.synthetic import fill_module_

    torch.manual_seed(0)
    m = {}
    m["denoising_unet"] = fill_module_(UNet3DConditionModel(**C.unet3d_kwargs(small)), seed, "denoising_unet.")
    m["reference_unet"] = fill_module_(UNet2DConditionModel(**C.unet2d_kwargs(small)), seed, "reference_unet.")
    m["vae"] = fill_module_(AutoencoderKL(**(C.SD_VAE_SMALL if small else C.SD_VAE_FT_MSE)), seed, "vae.")
    ch0 = (C.SD15_UNET_SMALL if small else C.SD15_UNET)["block_out_channels"][0]
    m["pose_guider"] = fill_module_(PoseGuider(noise_latent_channels=ch0, use_ca=True), seed, "pose_guider.")
    if with_clip:
        from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
        cfg = CLIPVisionConfig(**(C.CLIP_SMALL if small else C.CLIP_VIT_L14))
        m["image_encoder"] = fill_module_(CLIPVisionModelWithProjection(cfg), seed, "image_encoder.").eval()
    m["scheduler"] = DDIMScheduler(**C.DDIM_V2)
    return m
