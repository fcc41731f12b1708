"""`PoseGuider` (src/models/pose_guider.py:13-162) — SURVEY.md §8f rank 1: independent of the DDIM timestep,
so the pipeline evaluates it once per context window instead of once per step
(src/pipelines/pipeline_pose2vid_long.py:531-536 recomputes it every step).  On the GPU it runs on the HIP
engine (`engine.pose_guider_forward`: direct / implicit-GEMM convolutions, BatchNorm+ReLU, the four
self-attention blocks on the MFMA GEMM and attention kernels) — stock torch convs in fp16 fall into MIOpen's
`naive_conv_*` kernels here (0.8 s per 512x512 clip).  CPU tensors take the plain-PyTorch restatement below
(the reference keeps this module in torch; it is what the CPU tests and the oracle comparison use).

Same constructor, state-dict names (incl. BatchNorm buffers) and forward signature as the reference.
Quirks kept on purpose:
  * the scripts never call `.eval()` (`scripts/pose2vid.py:77`), so BatchNorm normalises with BATCH
    statistics over the (b f) frames of the window; `self.training` selects that here too.  Running
    statistics are not updated (inference has no use for them).
  * `ref_x` only feeds the `cross_attn*` blocks' second argument, which they ignore
    (`cross_attention_dim=None` => no attn2, src/models/attention.py:122-146), so it never influences
    the output and is not evaluated.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .params import build_tree, pose_guider_shapes, pose_guider_stacks


class PoseGuider(nn.Module):
    def __init__(self, noise_latent_channels=320, use_ca=True):
        super().__init__()
        self.use_ca = use_ca
        self.noise_latent_channels = noise_latent_channels
        params, buffers = pose_guider_shapes(noise_latent_channels, use_ca)
        build_tree(self, params, buffers)
        self._stacks = pose_guider_stacks(noise_latent_channels)
        object.__setattr__(self, "_packed", None)
        with torch.no_grad():
            self.scale.fill_(2.0)
            self.final_proj.weight.zero_()  # reference init (:119-122); checkpoints overwrite it

    @property
    def dtype(self):
        return self.scale.dtype

    @property
    def device(self):
        return self.scale.device

    @classmethod
    def from_pretrained(cls, pretrained_model_path):
        import os
        if not os.path.exists(pretrained_model_path):
            print(f"There is no model file in {pretrained_model_path}")
        state_dict = torch.load(pretrained_model_path, map_location="cpu", weights_only=True)
        model = cls(noise_latent_channels=320)
        model.load_state_dict(state_dict, strict=True)
        return model

    # -- packed weights for the HIP engine (rebuilt lazily whenever the parameters change) ----------------
    def _apply(self, fn, *a, **k):
        object.__setattr__(self, "_packed", None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        object.__setattr__(self, "_packed", None)
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def packed(self):
        from .engine import PackedNet
        _lib.load()
        if self._packed is None or self._packed.device != self.device:
            object.__setattr__(self, "_packed", PackedNet(self.state_dict(), self.device))
        return self._packed

    @torch.no_grad()
    def forward_nhwc(self, x):
        """x (N, H, W, 3) fp16 on the GPU -> 5 channels-last feature maps (N, h, w, C) (HIP engine)."""
        from . import engine
        return engine.pose_guider_forward(self.packed(), self._stacks, x, self.training, self.use_ca)

    # ------------------------------------------------------------------------------------------------
    def _p(self, name):
        return self.get_parameter(name)

    def _bn_relu(self, name, x):
        w, b = self._p(name + ".weight"), self._p(name + ".bias")
        if self.training:
            return F.relu(F.batch_norm(x, None, None, w, b, training=True, eps=1e-5))
        return F.relu(F.batch_norm(x, self.get_buffer(name + ".running_mean"), self.get_buffer(name + ".running_var"),
                                   w, b, training=False, eps=1e-5))

    def _stack(self, name, x):
        _cin, layers = self._stacks[name]
        for k, (_co, _ks, stride, pad) in enumerate(layers):
            x = F.conv2d(x, self._p(f"{name}.{3 * k}.weight"), self._p(f"{name}.{3 * k}.bias"), stride=stride,
                         padding=pad)
            x = self._bn_relu(f"{name}.{3 * k + 1}", x)
        return x

    def _self_attn_block(self, p, x):
        """pose_guider.Transformer2DModel (:165-308): GN -> 1x1 -> [LN, 16-head self-attention, +res,
        LN, GEGLU FF, +res] -> 1x1 -> +residual"""
        N, C, H, W = x.shape
        h = F.group_norm(x, 32, self._p(p + ".norm.weight"), self._p(p + ".norm.bias"), 1e-6)
        h = F.conv2d(h, self._p(p + ".proj_in.weight"), self._p(p + ".proj_in.bias"))
        inner = h.shape[1]
        h = h.permute(0, 2, 3, 1).reshape(N, H * W, inner)
        b = p + ".transformer_blocks.0"
        heads = 16
        n1 = F.layer_norm(h, (inner,), self._p(b + ".norm1.weight"), self._p(b + ".norm1.bias"))
        q = F.linear(n1, self._p(b + ".attn1.to_q.weight")).view(N, -1, heads, inner // heads).transpose(1, 2)
        k = F.linear(n1, self._p(b + ".attn1.to_k.weight")).view(N, -1, heads, inner // heads).transpose(1, 2)
        v = F.linear(n1, self._p(b + ".attn1.to_v.weight")).view(N, -1, heads, inner // heads).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(N, -1, inner)
        h = F.linear(a, self._p(b + ".attn1.to_out.0.weight"), self._p(b + ".attn1.to_out.0.bias")) + h
        n3 = F.layer_norm(h, (inner,), self._p(b + ".norm3.weight"), self._p(b + ".norm3.bias"))
        val, gate = F.linear(n3, self._p(b + ".ff.net.0.proj.weight"), self._p(b + ".ff.net.0.proj.bias")).chunk(2, -1)
        h = F.linear(val * F.gelu(gate), self._p(b + ".ff.net.2.weight"), self._p(b + ".ff.net.2.bias")) + h
        h = h.reshape(N, H, W, inner).permute(0, 3, 1, 2)
        return F.conv2d(h, self._p(p + ".proj_out.weight"), self._p(p + ".proj_out.bias")) + x

    @torch.no_grad()
    def forward(self, x, ref_x=None):
        """x (b, 3, f, H, W) -> 5 feature maps (b, C, f, h, w) at 1/8, 1/16, 1/32, 1/64, 1/64 resolution."""
        b, c, f, H, W = x.shape
        if x.is_cuda:
            from . import hipops as ops
            return [ops.nhwc_to_ncfhw(t, b, out_f32=(x.dtype == torch.float32))
                    for t in self.forward_nhwc(ops.ncfhw_to_nhwc(x))]
        x = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, H, W)

        def out(t):
            return t.reshape(b, f, *t.shape[1:]).permute(0, 2, 1, 3, 4)

        fea = []
        x = self._stack("conv_layers", x)
        x = F.conv2d(x, self.final_proj.weight, self.final_proj.bias) * self.scale
        fea.append(out(x))
        for i in range(1, 5):
            x = self._stack(f"conv_layers_{i}", x)
            if self.use_ca:
                x = self._self_attn_block(f"cross_attn{i}", x)
            fea.append(out(x))
        return fea
