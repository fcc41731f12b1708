"""TEST INFRASTRUCTURE — a plain-PyTorch CPU emulation of `aniportrait_amd.hipops` (one function per C-ABI
wrapper, same signatures, fp32 arithmetic, one rounding to fp16 per op like the kernels).

It exists so that the HOST logic above the C ABI (engine.py's walk of the UNets / VAE / PoseGuider, weight
packing, the pipelines' window / CFG / DDIM orchestration) can be checked against the CPU oracle in the
build container, which has no GPU.  It is never imported by the product: `install()` monkeypatches the
module attributes for the duration of a test.  The kernels themselves are checked on the GPU
(tests/test_hip_ops.py, tests/test_gpu_models.py)."""

import torch
import torch.nn.functional as F

F16, F32 = torch.float16, torch.float32


def groupnorm(x1, gamma, beta, groups, eps, silu, x2=None, frames_per_stat=1):
    x = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
    N, HW, C = x.shape
    f = int(frames_per_stat)
    xs = x.float().reshape(N // f, f * HW, C)          # statistics over f consecutive images
    y = F.group_norm(xs.permute(0, 2, 1), groups, gamma.float(), beta.float(), eps).permute(0, 2, 1).reshape(N, HW, C)
    if silu:
        y = F.silu(y)
    return y.to(F16).contiguous()


def rowgemm320_supported(M, C, rows_per_frame=0):
    return C == 320 and M > 0 and M % 128 == 0 and (rows_per_frame == 0 or rows_per_frame % 32 == 0)


def groupnorm_scale_shift(x, gamma, beta, groups, eps):
    N, HW, C = x.shape
    xs = x.float().reshape(N, HW, groups, C // groups)
    mean = xs.mean(dim=(1, 3))
    var = xs.var(dim=(1, 3), unbiased=False)
    rstd = (var + eps).rsqrt()
    sc = rstd.repeat_interleave(C // groups, dim=1) * gamma.float()[None]
    sh = beta.float()[None] - mean.repeat_interleave(C // groups, dim=1) * sc
    return torch.stack([sc, sh], dim=-1).contiguous()


def affine_linear320(x, scale_shift, rows_per_frame, W, bias=None):
    M, C = x.shape
    idx = torch.arange(M) // rows_per_frame
    xn = (x.float() * scale_shift[idx, :, 0] + scale_shift[idx, :, 1]).to(F16)
    y = xn.float() @ W.float().t()
    if bias is not None:
        y = y + bias.float()
    return y.to(F16)


def ln_qkv_projection(x, gamma, beta, w_qkv, heads, q_alpha, eps=1e-5):
    M, C = x.shape
    d = C // heads
    nh = layernorm(x, gamma, beta, eps).float()
    y = nh @ w_qkv.float().t()
    q = (q_alpha * y[:, :C]).to(F16)
    k = y[:, C:2 * C].to(F16).reshape(M, heads, d).permute(1, 0, 2).contiguous()
    vt = y[:, 2 * C:].to(F16).t().contiguous()
    return q, k, vt


def layernorm(x, gamma, beta, eps=1e-5, pe=None, rows_per_frame=0, frames=0):
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps)
    if pe is not None:
        idx = (torch.arange(x.shape[0]) // rows_per_frame) % frames
        y = y + pe.float()[idx]
    return y.to(F16)


def _geglu_unpack(y):
    """columns packed per 32 as [16 value | 16 gate] -> value * gelu(gate)"""
    M, N = y.shape
    t = y.reshape(M, N // 32, 2, 16)
    return (t[:, :, 0] * F.gelu(t[:, :, 1])).reshape(M, N // 2)


def gemm(A, W, bias=None, A2=None, rowbias=None, rows_per_group=0, residual=None, act=0, out_f32=False,
         alpha=1.0, out=None, conv=None, batch=1, ldo=None, ldr=None, trans_out=False, head_dim=0):
    assert A.dtype == F16 and W.dtype == F16
    if A.dim() == 3 or W.dim() == 3:
        y = alpha * torch.matmul(A.float(), W.float().transpose(-1, -2))
        if y.dim() == 2:
            y = y.unsqueeze(0).expand(batch, -1, -1)
        assert bias is None and residual is None and rowbias is None and act == 0
        return y.contiguous() if out_f32 else y.to(F16).contiguous()
    if conv is not None:
        N_, H, Wd, Cin = conv["Nimg"], conv["Hin"], conv["Win"], conv["Cin"]
        x = A.reshape(N_, H, Wd, Cin).float().permute(0, 3, 1, 2)
        if conv.get("upsample"):
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        from aniportrait_amd.hipops import unpack_conv3x3
        w = unpack_conv3x3(W.float(), Cin, conv.get("korder", 1))
        pad = conv["pad"]
        He, We = x.shape[2], x.shape[3]
        # high-side padding implied by Hout
        ph = (conv["Hout"] - 1) * conv["stride"] + 3 - He - pad
        pw = (conv["Wout"] - 1) * conv["stride"] + 3 - We - pad
        x = F.pad(x, (pad, max(pw, 0), pad, max(ph, 0)))
        y = F.conv2d(x, w, None, stride=conv["stride"])[:, :, : conv["Hout"], : conv["Wout"]]
        y = alpha * y.permute(0, 2, 3, 1).reshape(-1, W.shape[0])
    else:
        a = A.float() if A2 is None else torch.cat([A.float(), A2.float()], dim=1)
        y = alpha * (a @ W.float().t())
    M, N = y.shape
    if bias is not None:
        y = y + bias.float()
    if act == 1:
        assert residual is None and rowbias is None
        y = _geglu_unpack(y)
    if rowbias is not None:
        idx = torch.arange(M) // rows_per_group
        y = y + rowbias.float()[idx][:, :N]
    if act == 2:
        y = y * torch.sigmoid(1.702 * y)
    if residual is not None:
        y = y + residual.float().reshape(M, -1)
    if trans_out:
        y = y.t()
    if head_dim:
        y = y.reshape(M, N // head_dim, head_dim).permute(1, 0, 2)
    y = y.contiguous() if out_f32 else y.to(F16).contiguous()
    if out is not None:
        if trans_out and ldo is not None and ldo != M:        # transposed rows at a padded pitch: out (N, ldo >= M)
            out[:, :M].copy_(y)
        else:
            out.copy_(y.reshape(out.shape))
        return out
    return y


def ffn_geglu(x, w1p, b1p, w2, b2, residual=None):
    h = _geglu_unpack(x.float() @ w1p.float().t() + b1p.float()).to(F16)     # the kernel rounds H to fp16 too
    y = h.float() @ w2.float().t()
    if b2 is not None:
        y = y + b2.float()
    if residual is not None:
        y = y + residual.float()
    return y.to(F16)


def ffn_geglu_ln(x, gamma, beta, w1p, b1p, w2, b2, residual=None, eps=1e-5):
    return ffn_geglu(layernorm(x, gamma, beta, eps), w1p, b1p, w2, b2, residual)


def conv3x3(x, Wp, bias, stride=1, pad=1, upsample=False, pad_hi=None, rowbias=None, rows_per_group=0,
            residual=None, out_f32=False, korder=None):
    N, H, Wd, Cin = x.shape
    if pad_hi is None:
        pad_hi = pad
    He, We = (2 * H, 2 * Wd) if upsample else (H, Wd)
    Ho = (He + pad + pad_hi - 3) // stride + 1
    Wo = (We + pad + pad_hi - 3) // stride + 1
    from aniportrait_amd.hipops import conv_korder
    conv = dict(Nimg=N, Hin=H, Win=Wd, Cin=Cin, Hout=Ho, Wout=Wo, stride=stride, pad=pad, upsample=upsample,
                korder=conv_korder(Cin) if korder is None else korder)
    res2 = residual.reshape(-1, Wp.shape[0]) if residual is not None else None
    out = gemm(x, Wp, bias, rowbias=rowbias, rows_per_group=rows_per_group, residual=res2, conv=conv,
               out_f32=out_f32)
    return out.reshape(N, Ho, Wo, Wp.shape[0])


def conv_direct(x, wp, bias, Cout, ksize, stride=1, pad=1, relu=False, residual=None):
    Cin = x.shape[-1]
    w = wp.float()[:, :Cout].reshape(ksize, ksize, Cin, Cout).permute(3, 2, 0, 1)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w, None if bias is None else bias.float(), stride=stride, padding=pad)
    y = y.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual.float().reshape(y.shape)
    if relu:
        y = F.relu(y)
    return y.to(F16).contiguous()


def batchnorm(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, relu=True):
    train = running_mean is None
    y = F.batch_norm(x.float(), None if train else running_mean.float().clone(), None if train else running_var.float().clone(),
                     gamma.float(), beta.float(), training=train, eps=eps)
    if relu:
        y = F.relu(y)
    return y.to(F16)


def ref_attention(q, ldq, k, ldk, vt, ldvt, n_frames, T, heads, d, kref=None, ldkr=0, vtref=None, ldvtr=0,
                  ref_index=None, scale=None, n_ref_frames=0, k_head_stride=0, kref_head_stride=0, q_log2_scaled=False,
                  frame_mod=0):
    C = heads * d
    nsrc = frame_mod if frame_mod else n_frames        # frames held by q / k / vt
    scale = d ** -0.5 if scale is None else scale
    if q_log2_scaled:       # q carries scale * log2(e): softmax of 2^(q.k)
        scale = 0.6931471805599453
    Q = q[:, :C].float().reshape(nsrc, T, heads, d).permute(0, 2, 1, 3)
    if k_head_stride:       # head-major (heads, tokens, d)
        k = k.reshape(heads, -1, d).permute(1, 0, 2).reshape(-1, C)
    if kref is not None and kref_head_stride:
        kref = kref.reshape(heads, -1, d).permute(1, 0, 2).reshape(-1, C)
    K = k[:, :C].float().reshape(nsrc, T, heads, d).permute(0, 2, 1, 3)
    V = vt[:, :nsrc * T].float().t().reshape(nsrc, T, heads, d).permute(0, 2, 1, 3)     # (rows may be padded to a 16-B pitch)
    out = torch.empty((n_frames, T, C), dtype=F32)
    for n in range(n_frames):
        Kn, Vn = K[n % nsrc], V[n % nsrc]
        r = -1 if ref_index is None else int(ref_index[n])
        if r >= 0:
            Kr = kref[r * T:(r + 1) * T, :C].float().reshape(T, heads, d).permute(1, 0, 2)
            Vr = vtref.float().t()[r * T:(r + 1) * T].reshape(T, heads, d).permute(1, 0, 2)
            Kn, Vn = torch.cat([Kn, Kr], dim=1), torch.cat([Vn, Vr], dim=1)
        p = torch.softmax(Q[n % nsrc] @ Kn.transpose(-1, -2) * scale, dim=-1)
        out[n] = (p @ Vn).permute(1, 0, 2).reshape(T, C)
    return out.reshape(n_frames * T, C).to(F16)


def attn_q_alpha(d, scale=None):
    return float((d ** -0.5 if scale is None else scale) * 1.4426950408889634)


def temporal_attention(qkv, B, Fr, T, heads, d, scale=None):
    C = heads * d
    scale = d ** -0.5 if scale is None else scale
    x = qkv.float().reshape(B, Fr, T, 3, heads, d)
    q, k, v = (x[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))  # (B, T, heads, Fr, d)
    p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    o = (p @ v).permute(0, 3, 1, 2, 4)  # (B, Fr, T, heads, d)
    return o.reshape(B * Fr * T, C).to(F16).contiguous()


def temporal_qkv_attention_supported(Fr, T, C, heads):
    return Fr == 16 and C == 320 and heads == 8 and T > 0 and T % 8 == 0


def temporal_qkv_attention(x, gamma, beta_pe, w_packed, B, Fr, T, heads, eps=1e-5, scale=None):
    """LayerNorm + (beta + pe[frame]) -> packed [q | k | v per head pair] projection -> temporal attention, with the fp16
    roundings the kernel has (normalised rows, q / k / v)"""
    M, C = x.shape
    d = C // heads
    nh = F.layer_norm(x.float(), (C,), gamma.float(), None, eps)
    idx = (torch.arange(M) // T) % Fr
    nh = (nh + beta_pe.float()[idx]).to(F16)
    y = (nh.float() @ w_packed.float().t()).to(F16).reshape(M, 4, 3, 80)          # (row, head pair, q|k|v, 80)
    qkv = y.permute(0, 2, 1, 3).reshape(M, 3 * C).contiguous()                     # q | k | v, heads in order
    return temporal_attention(qkv, B, Fr, T, heads, d, scale)


def softmax_rows(s):
    return torch.softmax(s.float(), dim=-1).to(F16)


def linear_small(x, W, bias=None, silu_in=False, out=None):
    xx = F.silu(x.float()) if silu_in else x.float()
    y = xx @ W.float().t()
    y = y + bias.float() if bias is not None else y
    if out is not None:
        out.copy_(y)
        return out
    return y


def add(a, b):
    return (a.float() + b.float()).to(F16)


def window_accumulate(pred, acc, counter, frames, S, Fw, L_, HWC):
    idx = frames.long()
    keep = idx >= 0                      # masked slots (earlier duplicates of a wrapped dilated window) are skipped
    acc.view(S, L_, HWC)[:, idx[keep]] += pred.float().reshape(S, Fw, HWC)[:, keep]
    counter[idx[keep]] += 1.0


def cfg_ddim_step(acc, counter, latents, latents_f16, S, L_, HWC, guidance, sa, sb, sap, sbp):
    a = acc.view(S, L_, HWC)
    if S == 2:
        c = counter.view(L_, 1)
        u, cd = a[0] / c, a[1] / c
        v = u + guidance * (cd - u)
    else:
        v = a[0]
    x = latents.view(L_, HWC)
    x0 = sa * x - sb * v
    ep = sa * v + sb * x
    xn = sap * x0 + sbp * ep
    x.copy_(xn)
    if latents_f16 is not None:
        latents_f16.view(L_, HWC).copy_(xn.to(F16))


def ncfhw_to_nhwc(src):
    B, C, Fr, H, W = src.shape
    return src.permute(0, 2, 3, 4, 1).reshape(B * Fr, H, W, C).to(F16).contiguous()


def nhwc_to_ncfhw(src, B, out_f32=False, scale=1.0, shift=0.0, clamp01=False):
    BF, H, W, C = src.shape
    y = src.float().reshape(B, BF // B, H, W, C).permute(0, 4, 1, 2, 3) * scale + shift
    if clamp01:
        y = y.clamp(0.0, 1.0)
    return y.contiguous() if out_f32 else y.to(F16).contiguous()


def u8_to_f16(src, scale=1.0, shift=0.0):
    return (src.float() * scale + shift).to(F16)


def f16_to_u8(src, scale=1.0, shift=0.0):
    return ((src.float() * scale + shift).clamp(0, 1).to(F16).float() * 255.0).to(torch.uint8)


_EMULATED = ("groupnorm", "layernorm", "rowgemm320_supported", "groupnorm_scale_shift", "affine_linear320", "ln_qkv_projection", "gemm", "ffn_geglu", "ffn_geglu_ln", "conv3x3", "conv_direct", "batchnorm", "ref_attention",
             "temporal_attention", "temporal_qkv_attention", "temporal_qkv_attention_supported", "softmax_rows", "linear_small", "add", "window_accumulate", "cfg_ddim_step",
             "ncfhw_to_nhwc", "nhwc_to_ncfhw", "u8_to_f16", "f16_to_u8")


def install(monkeypatch):
    """Route aniportrait_amd.hipops through the emulation and let the HIP-backed modules pack their weights on
    the CPU (the product refuses to: `HipModel.packed()` raises off-GPU)."""
    from aniportrait_amd import clip_vision, engine, hipops, modeling, pipeline_pose2vid_long, pose_guider
    g = globals()
    for name in _EMULATED:
        assert hasattr(hipops, name), name
        monkeypatch.setattr(hipops, name, g[name])

    def packed(self):
        if self._packed is None:
            object.__setattr__(self, "_packed", engine.PackedNet(self.state_dict(), self.device))
        return self._packed

    monkeypatch.setattr(modeling.HipModel, "packed", packed)
    monkeypatch.setattr(pose_guider.PoseGuider, "packed", packed)
    monkeypatch.setattr(pipeline_pose2vid_long.Pose2VideoPipeline, "_require_gpu", staticmethod(lambda device: None))
    monkeypatch.setattr(clip_vision.CLIPVisionHip, "_require_gpu", staticmethod(lambda device: None))
