"""Pins the test-only restatement of `diffusers==0.24.0` (oracle/diffusers_stub/) to the REAL
library wherever one is importable: the golden fixtures were produced by the reference's own files running on that
restatement, so "parity unpinned at the diffusers boundary"  closes exactly where this file runs green
against a wheel.  (`/root/reference/requirements.txt:5` pins 0.24.0; neither the build container nor the GPU image
ships it — there the module skips and PRINTS why, so the record says by itself that the boundary stayed unpinned.)

CPU only, not `gpu`-marked: collected by `-m "not gpu"` here and importable on the GPU box.  The stub is loaded under
the private name `anip_diffusers_stub` (its modules use relative imports only), so it can sit beside a real
`diffusers` in one process.  Compared on seeded inputs, weights copied through `state_dict`, tolerance 1e-6 (fp32
against fp32, the same torch ops expected on both sides):
  Attention + AttnProcessor / AttnProcessor2_0 (self and cross), FeedForward / GEGLU, Timesteps / TimestepEmbedding,
  ResnetBlock2D, Downsample2D / Upsample2D, DDIMScheduler.set_timesteps / step (inference_v2 and inference_v1
  configs), VaeImageProcessor.preprocess (PIL and numpy paths), AutoencoderKL.encode / decode.
A constructor the wheel rejects (a newer diffusers with another signature) skips that case with the error text;
a NUMERICAL difference fails."""
import importlib
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(REPO, "oracle", "diffusers_stub", "diffusers")
TOL = 1e-6


def _real_diffusers():
    """the real package, or (None, reason): a `diffusers` that resolves into oracle/diffusers_stub is not the wheel.
    ANIP_STUB_SELFCHECK=1 (development only) accepts the stub itself as "the wheel": every comparison then runs stub
    against stub — a check of THIS file's code paths, not of the restatement."""
    if os.environ.get("ANIP_STUB_SELFCHECK") == "1":
        sys.path.insert(0, os.path.dirname(STUB_DIR))
        return importlib.import_module("diffusers"), None
    saved = list(sys.path)
    try:
        sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != os.path.dirname(STUB_DIR)]
        m = sys.modules.get("diffusers")
        if m is not None and os.path.abspath(getattr(m, "__file__", "") or "").startswith(os.path.dirname(STUB_DIR)):
            return None, "the `diffusers` already imported in this process is the test-only stub"
        try:
            spec = importlib.util.find_spec("diffusers")
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin:
            return None, "no `diffusers` wheel is installed in this image (offline; requirements.txt:5 pins 0.24.0)"
        if os.path.abspath(spec.origin).startswith(os.path.dirname(STUB_DIR)):
            return None, "`diffusers` resolves to oracle/diffusers_stub only"
        try:
            return importlib.import_module("diffusers"), None
        except Exception as e:  # a wheel that cannot be imported (missing dependency) is no oracle either
            return None, f"`diffusers` is installed but does not import: {type(e).__name__}: {e}"
    finally:
        sys.path[:] = saved


def _stub():
    name = "anip_diffusers_stub"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(STUB_DIR, "__init__.py"),
                                                  submodule_search_locations=[STUB_DIR])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


REAL, WHY_NOT = _real_diffusers()
if REAL is None:
    print(f"\n[test_stub_vs_diffusers] SKIPPED — parity stays UNPINNED at the diffusers boundary: {WHY_NOT}")
needs_wheel = pytest.mark.skipif(REAL is None, reason=f"diffusers boundary stays unpinned: {WHY_NOT}")


def _sub(path):
    """(stub module, real module) of e.g. 'models.attention_processor'"""
    _stub()
    return importlib.import_module("anip_diffusers_stub." + path), importlib.import_module("diffusers." + path)


def _make(cls, *a, **k):
    try:
        return cls(*a, **k)
    except TypeError as e:
        pytest.skip(f"the installed diffusers ({getattr(REAL, '__version__', '?')}) rejects the 0.24.0 constructor call: {e}")


def _pair(stub_cls, real_cls, *a, seed=0, **k):
    torch.manual_seed(seed)
    s = stub_cls(*a, **k).eval()
    r = _make(real_cls, *a, **k).eval()
    missing = r.load_state_dict(s.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return s, r


def _close(a, b, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
    assert err <= TOL, f"{what}: stub differs from diffusers {getattr(REAL, '__version__', '?')} by {err:.3e}"


def test_the_stub_loads_under_a_private_name_and_the_outcome_is_recorded():
    """runs everywhere: the stub is importable beside a real package, and the record of this run says whether the wheel
    comparison happened"""
    m = _stub()
    assert m.__name__ == "anip_diffusers_stub" and hasattr(m, "DDIMScheduler") and hasattr(m, "AutoencoderKL")
    sch = m.DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                          clip_sample=False, steps_offset=1, prediction_type="v_prediction",
                          rescale_betas_zero_snr=True, timestep_spacing="trailing")
    sch.set_timesteps(25)
    assert int(sch.timesteps[0]) == 999 and len(sch.timesteps) == 25
    print("[test_stub_vs_diffusers] wheel comparison " +
          ("RAN against diffusers " + str(getattr(REAL, "__version__", "?")) if REAL is not None else f"did NOT run: {WHY_NOT}"))


@needs_wheel
@pytest.mark.parametrize("proc", ["AttnProcessor", "AttnProcessor2_0"])
@pytest.mark.parametrize("cross", [False, True])
def test_attention_and_processors(proc, cross):
    S, R = _sub("models.attention_processor")
    kw = dict(query_dim=64, heads=4, dim_head=16, bias=False, upcast_attention=False,
              cross_attention_dim=48 if cross else None)
    s, r = _pair(S.Attention, R.Attention, **kw)
    s.set_processor(getattr(S, proc)())
    r.set_processor(getattr(R, proc)())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 37, 64, generator=g)
    e = torch.randn(2, 5, 48, generator=g) if cross else None
    with torch.no_grad():
        _close(s(x, encoder_hidden_states=e), r(x, encoder_hidden_states=e), f"Attention/{proc}/cross={cross}")


@needs_wheel
def test_feedforward_geglu():
    S, R = _sub("models.attention")
    s, r = _pair(S.FeedForward, R.FeedForward, 64, mult=4, activation_fn="geglu")
    x = torch.randn(3, 11, 64, generator=torch.Generator().manual_seed(2)) * 3
    with torch.no_grad():
        _close(s(x), r(x), "FeedForward(geglu)")


@needs_wheel
def test_timesteps_and_timestep_embedding():
    S, R = _sub("models.embeddings")
    t = torch.tensor([999, 958, 500, 1, 0])
    _close(S.Timesteps(320, True, 0)(t), _make(R.Timesteps, 320, True, 0)(t), "Timesteps(320, flip, shift 0)")
    s, r = _pair(S.TimestepEmbedding, R.TimestepEmbedding, 320, 1280)
    x = S.Timesteps(320, True, 0)(t)
    with torch.no_grad():
        _close(s(x), r(x), "TimestepEmbedding")


@needs_wheel
@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 128)])
def test_resnet_block_2d(cin, cout):
    S, R = _sub("models.resnet")
    kw = dict(in_channels=cin, out_channels=cout, temb_channels=96, groups=32, eps=1e-5, non_linearity="silu",
              output_scale_factor=1.0)
    s, r = _pair(S.ResnetBlock2D, R.ResnetBlock2D, **kw)
    g = torch.Generator().manual_seed(3)
    x, temb = torch.randn(2, cin, 12, 12, generator=g), torch.randn(2, 96, generator=g)
    with torch.no_grad():
        _close(s(x, temb), r(x, temb), f"ResnetBlock2D {cin}->{cout}")
    # the VAE's form: no time embedding
    s, r = _pair(S.ResnetBlock2D, R.ResnetBlock2D, in_channels=cin, out_channels=cout, temb_channels=None, groups=32,
                 eps=1e-6, non_linearity="silu")
    with torch.no_grad():
        _close(s(x, None), r(x, None), f"ResnetBlock2D (VAE form) {cin}->{cout}")


@needs_wheel
def test_down_and_upsample_2d():
    S, R = _sub("models.resnet")
    x = torch.randn(2, 32, 10, 10, generator=torch.Generator().manual_seed(4))
    s, r = _pair(S.Downsample2D, R.Downsample2D, 32, use_conv=True, out_channels=32, padding=1, name="op")
    with torch.no_grad():
        _close(s(x), r(x), "Downsample2D(padding=1)")
    s, r = _pair(S.Downsample2D, R.Downsample2D, 32, use_conv=True, out_channels=32, padding=0, name="op")
    with torch.no_grad():
        _close(s(x), r(x), "Downsample2D(padding=0: the VAE encoder's asymmetric pad)")
    s, r = _pair(S.Upsample2D, R.Upsample2D, 32, use_conv=True, out_channels=32)
    with torch.no_grad():
        _close(s(x), r(x), "Upsample2D")


DDIM_V2 = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False,
               steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
DDIM_V1 = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False,
               steps_offset=1)


@needs_wheel
@pytest.mark.parametrize("cfg,steps", [(DDIM_V2, 25), (DDIM_V2, 4), (DDIM_V1, 25), (DDIM_V1, 3)])
def test_ddim_scheduler(cfg, steps):
    """configs/inference/inference_v2.yaml:25-34 and inference_v1.yaml:18-23"""
    S, R = _sub("schedulers")
    s, r = S.DDIMScheduler(**cfg), _make(R.DDIMScheduler, **cfg)
    _close(s.alphas_cumprod, r.alphas_cumprod, "alphas_cumprod")
    s.set_timesteps(steps)
    r.set_timesteps(steps)
    assert s.timesteps.tolist() == r.timesteps.tolist()
    assert float(s.init_noise_sigma) == float(r.init_noise_sigma)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 3, 8, 8, generator=g)
    xs, xr = x.clone(), x.clone()
    for t in s.timesteps:
        m = torch.randn(x.shape, generator=g)
        xs = s.step(m, t, xs, eta=0.0).prev_sample
        xr = r.step(m, t, xr, eta=0.0).prev_sample
        _close(xs, xr, f"DDIM step t={int(t)}")


@needs_wheel
def test_vae_image_processor_preprocess():
    """PIL path (`pipeline_pose2vid_long.py:424-427`: the reference image) and numpy path (the scripts' pose arrays)"""
    import PIL.Image
    S, R = _sub("image_processor")
    rng = np.random.RandomState(6)
    img = PIL.Image.fromarray(rng.randint(0, 256, (70, 90, 3), dtype=np.uint8))
    for kw in (dict(vae_scale_factor=8, do_convert_rgb=True), dict(vae_scale_factor=8, do_convert_rgb=True, do_normalize=False)):
        s, r = S.VaeImageProcessor(**kw), _make(R.VaeImageProcessor, **kw)
        _close(s.preprocess(img, height=64, width=64), r.preprocess(img, height=64, width=64), f"preprocess(PIL) {kw}")
        arr = rng.rand(2, 40, 48, 3).astype(np.float32)
        _close(s.preprocess(arr, height=32, width=32), r.preprocess(arr, height=32, width=32), f"preprocess(numpy) {kw}")


@needs_wheel
def test_autoencoder_kl_encode_decode():
    S, R = _sub("models.autoencoder_kl")
    cfg = dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2,
               block_out_channels=(32, 64), layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=32,
               sample_size=32)
    s, r = _pair(S.AutoencoderKL, R.AutoencoderKL, **cfg)
    g = torch.Generator().manual_seed(7)
    x, z = torch.randn(1, 3, 32, 32, generator=g), torch.randn(2, 4, 16, 16, generator=g)
    with torch.no_grad():
        _close(s.encode(x).latent_dist.mean, r.encode(x).latent_dist.mean, "AutoencoderKL.encode mean")
        _close(s.decode(z).sample, r.decode(z).sample, "AutoencoderKL.decode")
