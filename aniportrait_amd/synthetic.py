"""Deterministic synthetic weights and inputs (there are no checkpoints: no network).

Weights are generated **per parameter name** — ``Generator(seed ^ crc32(name))`` — so the same
values are obtained for the reference's modules, the CPU oracle and the HIP modules regardless
of module construction order, and golden fixtures made in the build container stay valid on the
GPU box.  Distributions follow PyTorch's default inits: conv/linear weights and
their biases ~ U(±1/sqrt(fan_in)); norm gammas 1 + 0.1·U(±1), norm betas 0.1·U(±1) (non-trivial
on purpose).  Zero-initialised layers of the reference (`motion_module.py:72-75`,
`pose_guider.py:120-122`) receive ordinary values so every path is live.  Values are rounded to
fp16 and (for fp32 consumers) up-cast again, so the fp32 oracle and the fp16 HIP path see
bit-identical weights.
"""
import math
import zlib

import numpy as np
import torch


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(name, shape, seed=0, like=None):
    """One synthetic parameter (fp32, fp16-representable)."""
    shape = tuple(shape)
    g = _gen(name, seed)
    u = torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        t = u / math.sqrt(fan_in)
    elif leaf == "bias":
        if like is not None and like.dim() >= 2:
            t = u / math.sqrt(int(np.prod(like.shape[1:])))
        else:
            t = 0.1 * u
    elif leaf == "weight":
        t = 1.0 + 0.1 * u
    elif leaf == "scale":
        t = torch.full(shape, 1.5)
    else:
        t = 0.1 * u
    return t.half().float()


def synth_state_dict(shapes, seed=0, prefix=""):
    """shapes: mapping name -> shape (e.g. ``{k: v.shape for k, v in module.state_dict().items()}``).
    Returns name -> fp32 tensor.  Only call with *parameter* names (buffers keep their values)."""
    out = {}
    for name, shape in shapes.items():
        like = None
        if name.endswith(".bias"):
            w = name[: -len("bias")] + "weight"
            if w in shapes and len(tuple(shapes[w])) >= 2:
                like = torch.empty(tuple(shapes[w]), device="meta")
        out[name] = synth_tensor(prefix + name, shape, seed, like)
    return out


@torch.no_grad()
def fill_module_(module, seed=0, prefix=""):
    """In-place synthetic fill of every *parameter* of ``module`` (buffers untouched)."""
    shapes = {k: tuple(p.shape) for k, p in module.named_parameters()}
    sd = synth_state_dict(shapes, seed, prefix)
    for k, p in module.named_parameters():
        p.copy_(sd[k].to(p.dtype))
    return module


def synth_pose_frames(L, H, W, seed0=1234):
    """uint8 (L, H, W, 3) sparse coloured line drawings on black (stand-in for landmark renderings,
    `scripts/pose2vid.py:132-160`)."""
    from PIL import Image, ImageDraw

    out = np.zeros((L, H, W, 3), dtype=np.uint8)
    for i in range(L):
        rng = np.random.default_rng(seed0 + i)
        im = Image.new("RGB", (W, H), (0, 0, 0))
        d = ImageDraw.Draw(im)
        for _ in range(40):
            x0, x1 = rng.integers(0, W, 2)
            y0, y1 = rng.integers(0, H, 2)
            col = tuple(int(c) for c in rng.integers(32, 256, 3))
            d.line([(int(x0), int(y0)), (int(x1), int(y1))], fill=col, width=max(1, W // 128))
        out[i] = np.asarray(im)
    return out


def synth_ref_image(H, W, seed=1):
    """PIL RGB reference image (uniform-random uint8)."""
    from PIL import Image

    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8), "RGB")


def synth_latents(L, h, w, seed=42, channels=4):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn((1, channels, L, h, w), generator=g, dtype=torch.float32)


@torch.no_grad()
def fast_fill_(module, seed=0):
    """Random fill ON the module's device (bench.py: no checkpoints, values only need to be well-scaled):
    weights ~ N(0, 1/fan_in), norm gammas 1 + 0.1 N, biases / betas 0.1 N.  Not reproducible across
    devices; parity tests use the name-hash `fill_module_` instead."""
    dev = next(module.parameters()).device
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + seed)
    for name, p in module.named_parameters():
        leaf = name.rsplit(".", 1)[-1]
        r = torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32)
        if p.dim() >= 2:
            r.mul_(1.0 / math.sqrt(int(np.prod(p.shape[1:]))))
        elif leaf == "weight":
            r.mul_(0.1).add_(1.0)
        elif leaf == "scale":
            r.fill_(1.5)
        else:
            r.mul_(0.1)
        p.copy_(r.to(p.dtype))
    return module
