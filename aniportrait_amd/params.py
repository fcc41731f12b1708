"""State-dict format of the networks on the pose2vid hot path, derived from their configs.

The reference's checkpoints (`denoising_unet.pth`, `reference_unet.pth`, `pose_guider.pth`,
SD-1.5 `unet/`, sd-vae-ft-mse) are an on-disk format: parameter names and shapes must be accepted
unchanged (`scripts/pose2vid.py:91-100`, `src/models/unet_3d.py:582-673`).  This module restates that
format as pure functions config -> {name: shape}; `tests/test_params_format.py` checks them against
manifests dumped from the reference's own modules (tests/golden/shapes_*.json).

`ParamModule` is the generic container: a tree of `torch.nn.Module`s whose attribute paths spell
the reference's names, holding `nn.Parameter`s / buffers only.  The arithmetic lives in
`aniportrait_amd/engine.py` (HIP kernels), not in `forward()` methods of the leaves.
"""
import math
from collections import OrderedDict

import torch
from torch import nn

_DOWN_HAS_ATTN = (True, True, True, False)
_UP_HAS_ATTN = (False, True, True, True)


# ----------------------------------------------------------------------------------------------------
# name -> shape generators
# ----------------------------------------------------------------------------------------------------

def _norm(d, p, c):
    d[p + ".weight"] = (c,)
    d[p + ".bias"] = (c,)


def _conv(d, p, co, ci, k):
    d[p + ".weight"] = (co, ci, k, k)
    d[p + ".bias"] = (co,)


def _linear(d, p, co, ci, bias=True):
    d[p + ".weight"] = (co, ci)
    if bias:
        d[p + ".bias"] = (co,)


def _resnet(d, p, ci, co, temb):
    """ResnetBlock3D / diffusers ResnetBlock2D (src/models/resnet.py:124-216)."""
    _norm(d, p + ".norm1", ci)
    _conv(d, p + ".conv1", co, ci, 3)
    if temb:
        _linear(d, p + ".time_emb_proj", co, temb)
    _norm(d, p + ".norm2", co)
    _conv(d, p + ".conv2", co, co, 3)
    if ci != co:
        _conv(d, p + ".conv_shortcut", co, ci, 1)


def _ff(d, p, c):
    _linear(d, p + ".net.0.proj", 8 * c, c)
    _linear(d, p + ".net.2", c, 4 * c)


def _attention(d, p, c, ctx=None, qkv_bias=False):
    _linear(d, p + ".to_q", c, c, qkv_bias)
    _linear(d, p + ".to_k", c, ctx or c, qkv_bias)
    _linear(d, p + ".to_v", c, ctx or c, qkv_bias)
    _linear(d, p + ".to_out.0", c, c)


def _spatial_transformer(d, p, c, cross_dim, inner=None):
    """Transformer3DModel / Transformer2DModel with one (Temporal)BasicTransformerBlock
    (src/models/transformer_3d.py:27-101, src/models/attention.py:300-381)."""
    inner = inner or c
    _norm(d, p + ".norm", c)
    _conv(d, p + ".proj_in", inner, c, 1)
    b = p + ".transformer_blocks.0"
    _norm(d, b + ".norm1", inner)
    _attention(d, b + ".attn1", inner)
    if cross_dim:
        _norm(d, b + ".norm2", inner)
        _attention(d, b + ".attn2", inner, cross_dim)
    _norm(d, b + ".norm3", inner)
    _ff(d, b + ".ff", inner)
    _conv(d, p + ".proj_out", c, inner, 1)


def _motion_module(d, bufs, p, c, n_attn, max_len):
    """VanillaTemporalModule (src/models/motion_module.py:44-259)."""
    p = p + ".temporal_transformer"
    _norm(d, p + ".norm", c)
    _linear(d, p + ".proj_in", c, c)
    b = p + ".transformer_blocks.0"
    for i in range(n_attn):
        _attention(d, b + f".attention_blocks.{i}", c)
        bufs[b + f".attention_blocks.{i}.pos_encoder.pe"] = (1, max_len, c)
    for i in range(n_attn):
        _norm(d, b + f".norms.{i}", c)
    _ff(d, b + ".ff", c)
    _norm(d, b + ".ff_norm", c)
    _linear(d, p + ".proj_out", c, c)


def unet_shapes(cfg, three_d):
    """(params, buffers) of UNet3DConditionModel (three_d) or the ReferenceNet UNet2DConditionModel
    with conv_norm_out/conv_out removed (src/models/unet_2d_condition.py:645-653)."""
    boc = tuple(cfg["block_out_channels"])
    lpb = cfg["layers_per_block"]
    cross = cfg["cross_attention_dim"]
    temb = boc[0] * 4
    mm = three_d and cfg.get("use_motion_module", False)
    mmk = cfg.get("motion_module_kwargs", {}) or {}
    n_attn = len(mmk.get("attention_block_types", ("Temporal_Self", "Temporal_Self")))
    max_len = mmk.get("temporal_position_encoding_max_len", 24)
    mm_res = tuple(cfg.get("motion_module_resolutions", (1, 2, 4, 8)))
    d, bufs = OrderedDict(), OrderedDict()

    def add_mm(p, c, res_idx):
        if mm and (res_idx is None or (2 ** res_idx) in mm_res):
            _motion_module(d, bufs, p, c, n_attn, max_len)

    _conv(d, "conv_in", boc[0], cfg["in_channels"], 3)
    _linear(d, "time_embedding.linear_1", temb, boc[0])
    _linear(d, "time_embedding.linear_2", temb, temb)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(lpb):
            _resnet(d, f"down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, temb)
        if _DOWN_HAS_ATTN[i]:
            for j in range(lpb):
                _spatial_transformer(d, f"down_blocks.{i}.attentions.{j}", co, cross)
        for j in range(lpb):
            add_mm(f"down_blocks.{i}.motion_modules.{j}", co, i)
        if i != len(boc) - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", co, co, 3)
        ch = co
    _resnet(d, "mid_block.resnets.0", ch, ch, temb)
    _resnet(d, "mid_block.resnets.1", ch, ch, temb)
    _spatial_transformer(d, "mid_block.attentions.0", ch, cross)
    if mm and cfg.get("motion_module_mid_block", False):
        _motion_module(d, bufs, "mid_block.motion_modules.0", ch, n_attn, max_len)
    rev = tuple(reversed(boc))
    prev = rev[0]
    for i, co in enumerate(rev):
        skip_in = rev[min(i + 1, len(boc) - 1)]
        for j in range(lpb + 1):
            res_skip = skip_in if j == lpb else co
            res_in = prev if j == 0 else co
            _resnet(d, f"up_blocks.{i}.resnets.{j}", res_in + res_skip, co, temb)
        if _UP_HAS_ATTN[i]:
            for j in range(lpb + 1):
                _spatial_transformer(d, f"up_blocks.{i}.attentions.{j}", co, cross)
        for j in range(lpb + 1):
            add_mm(f"up_blocks.{i}.motion_modules.{j}", co, len(boc) - 1 - i)
        if i != len(boc) - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    if three_d:
        _norm(d, "conv_norm_out", boc[0])
        _conv(d, "conv_out", cfg["out_channels"], boc[0], 3)
    return d, bufs


def vae_shapes(cfg):
    """diffusers AutoencoderKL (sd-vae-ft-mse topology)."""
    boc = tuple(cfg["block_out_channels"])
    lpb = cfg["layers_per_block"]
    lat = cfg["latent_channels"]
    d = OrderedDict()

    def mid(p, c):
        a = p + ".attentions.0"
        _norm(d, a + ".group_norm", c)
        _attention(d, a, c, qkv_bias=True)
        _resnet(d, p + ".resnets.0", c, c, 0)
        _resnet(d, p + ".resnets.1", c, c, 0)

    _conv(d, "encoder.conv_in", boc[0], cfg["in_channels"], 3)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(lpb):
            _resnet(d, f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, 0)
        if i != len(boc) - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
        ch = co
    mid("encoder.mid_block", ch)
    _norm(d, "encoder.conv_norm_out", ch)
    _conv(d, "encoder.conv_out", 2 * lat, ch, 3)
    rev = tuple(reversed(boc))
    _conv(d, "decoder.conv_in", rev[0], lat, 3)
    ch = rev[0]
    for i, co in enumerate(rev):
        for j in range(lpb + 1):
            _resnet(d, f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, 0)
        if i != len(boc) - 1:
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        ch = co
    mid("decoder.mid_block", rev[0])
    _norm(d, "decoder.conv_norm_out", ch)
    _conv(d, "decoder.conv_out", cfg["out_channels"], ch, 3)
    _conv(d, "quant_conv", 2 * lat, 2 * lat, 1)
    _conv(d, "post_quant_conv", lat, lat, 1)
    return d, OrderedDict()


# PoseGuider conv stacks: (out_channels, kernel, stride, padding) per Conv+BN+ReLU triple
# (src/models/pose_guider.py:17-118)
def pose_guider_stacks(noise_latent_channels=320):
    c = noise_latent_channels
    return OrderedDict([
        ("conv_layers", (3, [(3, 3, 1, 1), (16, 4, 2, 1), (16, 3, 1, 1), (32, 4, 2, 1), (32, 3, 1, 1),
                             (64, 4, 2, 1), (64, 3, 1, 1), (128, 3, 1, 1)])),
        ("conv_layers_1", (c, [(c, 3, 1, 1), (c, 3, 2, 1)])),
        ("conv_layers_2", (c, [(c, 3, 1, 1), (2 * c, 3, 2, 1)])),
        ("conv_layers_3", (2 * c, [(2 * c, 3, 1, 1), (4 * c, 3, 2, 1)])),
        ("conv_layers_4", (4 * c, [(4 * c, 3, 1, 1)])),
    ])


def pose_guider_shapes(noise_latent_channels=320, use_ca=True):
    c = noise_latent_channels
    d, bufs = OrderedDict(), OrderedDict()
    d["scale"] = (1,)
    stacks = pose_guider_stacks(c)

    def stack(name):
        ci, layers = stacks[name]
        for k, (co, ks, _s, _p) in enumerate(layers):
            _conv(d, f"{name}.{3 * k}", co, ci, ks)
            _norm(d, f"{name}.{3 * k + 1}", co)
            bufs[f"{name}.{3 * k + 1}.running_mean"] = (co,)
            bufs[f"{name}.{3 * k + 1}.running_var"] = (co,)
            bufs[f"{name}.{3 * k + 1}.num_batches_tracked"] = ()
            ci = co

    stack("conv_layers")
    _conv(d, "final_proj", c, 128, 1)
    for i in range(1, 5):
        stack(f"conv_layers_{i}")
    if use_ca:
        for i, ch in zip(range(1, 5), (c, 2 * c, 4 * c, 4 * c)):
            _spatial_transformer(d, f"cross_attn{i}", ch, None, inner=16 * 88)
    return d, bufs


# ----------------------------------------------------------------------------------------------------
# generic parameter tree
# ----------------------------------------------------------------------------------------------------

class ParamNode(nn.Module):
    """Inner node / leaf holder of a ParamModule tree (no forward: containers only)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("ParamNode is a parameter container; the arithmetic runs in aniportrait_amd.engine")


def sinusoidal_pe(d_model, max_len):
    """PositionalEncoding buffer (src/models/motion_module.py:262-277)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


_SKIP_INIT = False


class skip_init:
    """`with skip_init(): Model(...)` leaves parameters uninitialised (checkpoint / synthetic fill follows)."""

    def __enter__(self):
        global _SKIP_INIT
        self.prev, _SKIP_INIT = _SKIP_INIT, True

    def __exit__(self, *a):
        global _SKIP_INIT
        _SKIP_INIT = self.prev


def _default_init(name, shape):
    """PyTorch-default-like init (values only matter until a checkpoint is loaded)."""
    if _SKIP_INIT:
        return torch.empty(shape)
    leaf = name.rsplit(".", 1)[-1]
    # layers the reference zero-initialises (src/models/motion_module.py:72-75 `zero_module(proj_out)`,
    # src/models/pose_guider.py:120-122 `final_proj`): observable through `from_pretrained_2d(mm_zero_proj_out=True)`,
    # which leaves them at their constructor values
    if ".temporal_transformer.proj_out." in name or name.startswith("final_proj."):
        return torch.zeros(shape)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return (torch.rand(shape) * 2 - 1) / math.sqrt(fan_in)
    if leaf == "weight" or leaf == "scale":
        return torch.ones(shape)
    return torch.zeros(shape)


def build_tree(root, params, buffers):
    """Attach parameters/buffers to `root` under nested ParamNode children spelling their names."""
    def holder(path):
        m = root
        for part in path:
            if part not in m._modules:
                m.add_module(part, ParamNode())
            m = m._modules[part]
        return m

    for name, shape in params.items():
        *path, leaf = name.split(".")
        holder(path).register_parameter(leaf, nn.Parameter(_default_init(name, tuple(shape)), requires_grad=False))
    for name, shape in buffers.items():
        *path, leaf = name.split(".")
        if leaf == "pe":
            val = sinusoidal_pe(shape[2], shape[1])
        elif leaf == "running_var":
            val = torch.ones(tuple(shape))
        elif leaf == "num_batches_tracked":
            val = torch.zeros((), dtype=torch.long)
        else:
            val = torch.zeros(tuple(shape))
        holder(path).register_buffer(leaf, val)
    return root


def transformer_block_paths(params):
    """Structural paths of the spatial transformer blocks, in the order ReferenceAttentionControl
    pairs them: DFS (down -> up -> mid) stably sorted by -norm1 width
    (src/models/mutual_self_attention.py:321-337)."""
    paths = []
    for name in params:
        if name.endswith(".transformer_blocks.0.norm1.weight") and ".motion_modules." not in name:
            paths.append(name[: -len(".norm1.weight")])

    def dfs_key(p):
        top = p.split(".")[0]
        return {"down_blocks": 0, "up_blocks": 1, "mid_block": 2}[top]

    ordered = sorted(paths, key=dfs_key)  # stable: keeps in-block order
    return sorted(ordered, key=lambda p: -params[p + ".norm1.weight"][0])
