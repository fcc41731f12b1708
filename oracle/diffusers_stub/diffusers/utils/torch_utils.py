"""Test-only stub: `randn_tensor` restated from diffusers 0.24.0 (CPU generator => sample on
CPU in the requested dtype, then move), `apply_freeu` unused."""
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    rand_device = device
    device = device or torch.device("cpu")
    layout = layout or torch.strided
    if generator is not None:
        gen_device_type = generator.device.type if not isinstance(generator, list) else generator[0].device.type
        if gen_device_type != torch.device(device).type and gen_device_type == "cpu":
            rand_device = "cpu"
    if isinstance(generator, list):
        shape = (1,) + tuple(shape[1:])
        latents = torch.cat([torch.randn(shape, generator=g, device=rand_device, dtype=dtype, layout=layout)
                             for g in generator], dim=0).to(device)
    else:
        latents = torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype,
                              layout=layout).to(device)
    return latents


def apply_freeu(*a, **k):
    raise NotImplementedError("FreeU is never enabled on the hot path")
