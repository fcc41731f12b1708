"""Execution engine of the pose2vid hot path: walks the SD-1.5-topology UNets and the VAE with the HIP
kernels of libaniportrait_hip.so (through `hipops`).  No torch arithmetic on activations here —
torch owns memory and streams only.

Layout: every activation is channels-last fp16, an image batch (N, H, W, C) == token matrix
(N*H*W, C); N = b*f frames ordered (b f) as in the reference's `rearrange "b c f h w -> (b f) c h w"`
(src/models/resnet.py:14, src/models/transformer_3d.py:115), so the conv<->transformer permutes of
src/models/transformer_3d.py:128-153 do not exist here and every 1x1 conv is a plain GEMM.

`PackedNet` turns a reference-format state-dict into kernel-ready device tensors once:
3x3 conv weights [Cout][ky][kx][Cin], fused [to_q;to_k] / [to_q;to_k;to_v] projection matrices,
GEGLU rows interleaved per 16 output columns, all time_emb_proj layers stacked into one matrix, fp32 norm
parameters and biases.

Legal shortcuts relative to the reference's op sequence (identical results):
  * reference K/V (to_k/to_v of the ReferenceNet bank) are projected once per bank update, not per
    frame per step (src/models/mutual_self_attention.py:147-165);
  * CFG-unconditional frames attend to their own tokens only, in the same launch (:166-186);
  * attn2 over the length-1 CLIP sequence is softmax==1, i.e. the per-sample vector
    to_out(to_v(e)) added in attn1's output-projection epilogue (:191-205);
  * skip-concat / nearest-2x upsample / time-embedding add / residual adds are fused into the
    neighbouring GroupNorm / conv / GEMM kernels.
"""
import itertools
import math
import os

import torch

from . import hipops as ops

F16, F32 = torch.float16, torch.float32
_DOWN_HAS_ATTN = (True, True, True, False)
_UP_HAS_ATTN = (False, True, True, True)


# ----------------------------------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------------------------------

class PackedNet:
    """Kernel-ready weights of one network on one device.  `sd`: reference-format state-dict."""

    _serials = itertools.count(1)

    def __init__(self, sd, device):
        self.device = torch.device(device)
        self.sd = sd
        self.t = {}
        # process-unique, never reused: captured hipGraphs are keyed on it (`id()` of a freed PackedNet can be handed to
        # its replacement by the allocator, which would let a stale graph replay against freed weight addresses)
        self.serial = next(PackedNet._serials)

    # -- accessors (packed lazily, cached) ---------------------------------------------------------
    def _raw(self, name):
        return self.sd[name].detach()

    def has(self, name):
        return name in self.sd

    def f32(self, name):
        k = ("f32", name)
        if k not in self.t:
            self.t[k] = self._raw(name).to(self.device, F32).contiguous()
        return self.t[k]

    def opt_f32(self, name):
        return self.f32(name) if name in self.sd else None

    def lin(self, name):
        """Linear / 1x1-conv weight -> fp16 [N][K]."""
        k = ("lin", name)
        if k not in self.t:
            w = self._raw(name)
            self.t[k] = w.reshape(w.shape[0], -1).to(self.device, F16).contiguous()
        return self.t[k]

    def conv3(self, name):
        """3x3 conv weight -> fp16 [Cout][9*Cin] (tap-major)."""
        k = ("conv3", name)
        if k not in self.t:
            self.t[k] = ops.pack_conv3x3(self._raw(name).to(self.device, F16))
        return self.t[k]

    def conv_direct(self, name):
        """small-channel conv weight -> fp16 [k*k*Cin][Cout8] (LDS image of anip_conv_direct)."""
        k = ("convd", name)
        if k not in self.t:
            self.t[k] = ops.pack_conv_direct(self._raw(name).to(self.device, F16))
        return self.t[k]

    def scalar(self, name):
        """a one-element parameter as a Python float, read back once (a device-to-host copy per call would synchronise
        every forward and is not allowed while a stream is capturing)"""
        k = ("scalar", name)
        if k not in self.t:
            self.t[k] = float(self._raw(name).float().reshape(-1)[0])
        return self.t[k]

    def scaled_f32(self, name, scale):
        k = ("sf32", name, float(scale))
        if k not in self.t:
            self.t[k] = (self._raw(name).to(self.device, F32) * float(scale)).contiguous()
        return self.t[k]

    def cat_lin(self, names):
        k = ("cat",) + tuple(names)
        if k not in self.t:
            self.t[k] = torch.cat([self._raw(n).to(self.device, F16) for n in names], dim=0).contiguous()
        return self.t[k]

    def geglu(self, prefix):
        k = ("geglu", prefix)
        if k not in self.t:
            w = self._raw(prefix + ".weight").to(self.device, F16)
            b = self._raw(prefix + ".bias").to(self.device, F32)
            self.t[k] = ops.pack_geglu(w, b)
        return self.t[k]

    def pe(self, name, C):
        """positional-encoding table fp32 [max_len][C]"""
        k = ("pe", name)
        if k not in self.t:
            self.t[k] = self._raw(name).to(self.device, F32).reshape(-1, C).contiguous()
        return self.t[k]

    def beta_pe(self, norm, pe_name, C, f):
        """fp32 [f][C]: the LayerNorm's bias plus the frame's positional encoding (bias alone when the attention has no encoder)"""
        k = ("betape", norm, pe_name, int(f))
        if k not in self.t:
            t = self.f32(norm + ".bias")[None, :].repeat(f, 1)
            if pe_name is not None:
                t = t + self.pe(pe_name, C)[:f]
            self.t[k] = t.contiguous()
        return self.t[k]

    def tqkv(self, names):
        """[to_q; to_k; to_v] in anip_temporal_qkv_attention's head-pair row order"""
        k = ("tqkv",) + tuple(names)
        if k not in self.t:
            self.t[k] = ops.pack_temporal_qkv(*[self._raw(n).to(self.device, F16) for n in names])
        return self.t[k]

    def temb_stack(self, names):
        """all time_emb_proj layers stacked: W fp16 [sum Cout][temb], bias fp32, and column offsets."""
        k = ("tembstack",)
        if k not in self.t:
            W = torch.cat([self._raw(n + ".weight").to(self.device, F16) for n in names], dim=0).contiguous()
            b = torch.cat([self._raw(n + ".bias").to(self.device, F32) for n in names], dim=0).contiguous()
            offs, o = {}, 0
            for n in names:
                c = self.sd[n + ".weight"].shape[0]
                offs[n] = (o, c)
                o += c
            self.t[k] = (W, b, offs)
        return self.t[k]

    def vae_attn_out(self, a):
        """to_out weight + effective bias of the VAE mid attention:  softmax rows sum to 1, so the
        value bias commutes: P (V + 1 b_v^T) W_o^T + b_o = P V W_o^T + (W_o b_v + b_o)."""
        k = ("vaeo", a)
        if k not in self.t:
            Wo = self._raw(a + ".to_out.0.weight").to(self.device, F32)
            bo = self._raw(a + ".to_out.0.bias").to(self.device, F32)
            bv = self._raw(a + ".to_v.bias").to(self.device, F32)
            # the kernel sees fp16 weights; fold the bias with the same rounded matrix
            self.t[k] = (bo + Wo.to(F16).to(F32) @ bv).contiguous()
        return self.t[k]


# ----------------------------------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------------------------------

def resnet(net, p, x, skip, temb_rb, rows_per_group, eps, groups=32, gn_frames=1):
    """ResnetBlock3D / ResnetBlock2D (src/models/resnet.py:218-248) on x (N,H,W,C1) [+ skip (N,H,W,C2),
    the torch.cat of src/models/unet_3d_blocks.py:697,826 fused into norm1 and the shortcut GEMM].
    gn_frames: frames per GroupNorm statistic (1: InflatedGroupNorm, per frame; f: nn.GroupNorm over the 5-D tensor —
    use_inflated_groupnorm=False, src/models/resnet.py:161-164)."""
    N, H, W, C1 = x.shape
    HW = H * W
    x2 = None if skip is None else skip.reshape(N, HW, -1)
    h = ops.groupnorm(x.reshape(N, HW, C1), net.f32(p + ".norm1.weight"), net.f32(p + ".norm1.bias"), groups, eps,
                      True, x2=x2, frames_per_stat=gn_frames)
    Cin = h.shape[-1]
    h = ops.conv3x3(h.reshape(N, H, W, Cin), net.conv3(p + ".conv1.weight"), net.f32(p + ".conv1.bias"),
                    rowbias=temb_rb, rows_per_group=rows_per_group)
    Co = h.shape[-1]
    h = ops.groupnorm(h.reshape(N, HW, Co), net.f32(p + ".norm2.weight"), net.f32(p + ".norm2.bias"), groups, eps,
                      True, frames_per_stat=gn_frames)
    if net.has(p + ".conv_shortcut.weight"):
        sc = ops.gemm(x.reshape(N * HW, C1), net.lin(p + ".conv_shortcut.weight"), net.f32(p + ".conv_shortcut.bias"),
                      A2=None if skip is None else skip.reshape(N * HW, -1))
    else:
        assert skip is None
        sc = x.reshape(N * HW, C1)
    return ops.conv3x3(h.reshape(N, H, W, Co), net.conv3(p + ".conv2.weight"), net.f32(p + ".conv2.bias"),
                       residual=sc)


# single-kernel feed-forward at C = 320 (csrc/ffn.hip): FF1 + GEGLU + FF2 + residual without the M x 4C intermediate
# leaving the CU (335 MB written and read back per 64x64 layer otherwise); bit-identical to the two-GEMM path and, since
# round 2 (every step inside the captured graph), 1-4 % faster end to end (10.39 / 10.41 vs 9.99 / 10.31 frames/s in two
# paired runs).  ANIP_FUSED_FFN=0 selects the two-GEMM path.
_FUSED_FFN = os.environ.get("ANIP_FUSED_FFN", "1") == "1"
# the block's LayerNorm (norm3 / ff_norm) inside the fused kernel's prologue (round 5): no layernorm launch, one input tensor
# instead of two.  ANIP_FFN_LN=0 selects layernorm + anip_ffn_geglu (A/B measurements).
_FFN_LN = os.environ.get("ANIP_FFN_LN", "1") == "1"


# LayerNorm(+pe) -> to_q / to_k / to_v -> temporal attention as ONE launch at C = 320, F = 16 (csrc/tblock.hip): the normalised
# rows and the M x 960 q|k|v matrix never reach HBM.  ANIP_FUSED_TEMPORAL=0 selects the three-launch path (A/B measurements).
_FUSED_TEMPORAL = os.environ.get("ANIP_FUSED_TEMPORAL", "1") == "1"
# row-stationary projections at C = 320 (csrc/tblock.hip): norm1 -> to_q | to_k | to_v^T as one launch, and GroupNorm's apply
# inside proj_in (statistics finalised into a per-frame affine table).  ANIP_FUSED_ROWS=0 selects the separate launches.
_FUSED_ROWS = os.environ.get("ANIP_FUSED_ROWS", "1") == "1"
# unet_forward(cfg_shared_input=True): the CFG pair's identical prefix computed once.  ANIP_SHARE_CFG_PREFIX=0: both halves.
_SHARE_CFG_PREFIX = os.environ.get("ANIP_SHARE_CFG_PREFIX", "1") == "1"
_CHECK_CFG_PREFIX = os.environ.get("ANIP_CHECK_CFG_PREFIX", "0") == "1"


def transformer_in(net, p, x):
    """norm (GroupNorm 32, eps 1e-6, no activation) -> proj_in of Transformer3DModel / the temporal transformer
    (src/models/transformer_3d.py:128-139, src/models/motion_module.py:185-204; also the PoseGuider's Transformer2DModel,
    whose proj_in widens 320 -> 16 x 88 and therefore stays on the two launches): x (N, H, W, C) -> h (N*H*W, inner)"""
    N, H, W, C = x.shape
    T = H * W
    if _FUSED_ROWS and net.lin(p + ".proj_in.weight").shape[0] == C and ops.rowgemm320_supported(N * T, C, T):
        sst = ops.groupnorm_scale_shift(x.reshape(N, T, C), net.f32(p + ".norm.weight"), net.f32(p + ".norm.bias"), 32, 1e-6)
        return ops.affine_linear320(x.reshape(N * T, C), sst, T, net.lin(p + ".proj_in.weight"), net.f32(p + ".proj_in.bias"))
    h = ops.groupnorm(x.reshape(N, T, C), net.f32(p + ".norm.weight"), net.f32(p + ".norm.bias"), 32, 1e-6, False)
    return ops.gemm(h.reshape(N * T, C), net.lin(p + ".proj_in.weight"), net.f32(p + ".proj_in.bias"))


def feed_forward(net, p, h, norm):
    """LayerNorm `norm` -> diffusers FeedForward(geglu) -> + h (src/models/attention.py:361,436-445,
    src/models/motion_module.py:233-234,256-257): GEGLU fused in the first GEMM's epilogue."""
    M, C = h.shape
    fused = _FUSED_FFN and C == 320 and net.has(p + ".net.2.bias")
    wp, bp = net.geglu(p + ".net.0.proj")
    if fused and _FFN_LN:
        return ops.ffn_geglu_ln(h, net.f32(norm + ".weight"), net.f32(norm + ".bias"), wp, bp, net.lin(p + ".net.2.weight"),
                                net.f32(p + ".net.2.bias"), h)
    n_in = ops.layernorm(h, net.f32(norm + ".weight"), net.f32(norm + ".bias"))
    if fused:
        return ops.ffn_geglu(n_in, wp, bp, net.lin(p + ".net.2.weight"), net.f32(p + ".net.2.bias"), h)
    g = ops.gemm(n_in, wp, bp, act=1)
    return ops.gemm(g, net.lin(p + ".net.2.weight"), net.f32(p + ".net.2.bias"), residual=h)


class RefState:
    """Reference-attention state of one spatial transformer block (what `module.bank` plus the hacked
    forward's closure hold in src/models/mutual_self_attention.py:93-265)."""
    __slots__ = ("mode", "bank", "kref", "vtref", "stale", "written", "pool")

    def __init__(self):
        self.mode = "plain"   # "plain" | "write" | "read"
        self.bank = None      # (b, T, C) fp16 on device (read mode)
        self.kref = None      # (b*T, C) fp16: to_k(bank)
        self.vtref = None     # (C, b*T) fp16: to_v(bank)^T
        self.stale = True     # bank replaced since kref / vtref were projected: re-project IN PLACE
        self.written = None   # (b, T, C) fp16 produced in write mode
        # {(rows, C, device): (kref, vtref)} — one projection buffer pair per bank shape, allocated once and never
        # freed or re-allocated while the packed weights live: captured hipGraphs of the forward have these addresses
        # baked in, so a clip at another resolution (or a direct forward() call) in between must not recycle them
        self.pool = {}

    def buffers(self, rows, C, device):
        key = (int(rows), int(C), str(device))
        if key not in self.pool:
            self.pool[key] = (torch.empty((rows, C), dtype=F16, device=device),
                              torch.empty((C, rows), dtype=F16, device=device))
        return self.pool[key]

    def drop(self):
        self.bank = self.kref = self.vtref = None
        self.stale = True
        self.pool = {}


def project_reference_bank(net, p, ref, d):
    """K_ref = to_k(bank), V_ref^T = to_v(bank)^T of one hooked block (transformer-block path p), written IN PLACE into
    the block's per-shape buffers (src/models/mutual_self_attention.py:147-165 does this projection for every frame of
    every step; the bank is constant over a clip).  d = head dim: K_ref is stored head-major."""
    C = ref.bank.shape[-1]
    bank2 = ref.bank.reshape(-1, C)
    if bank2.dtype != F16 or bank2.device != net.device:
        bank2 = bank2.to(net.device, F16)
    kbuf, vbuf = ref.buffers(bank2.shape[0], C, net.device)
    ref.kref = ops.gemm(bank2, net.lin(p + ".attn1.to_k.weight"), out=kbuf, head_dim=d)   # head-major, like the frame's own K
    ref.vtref = ops.gemm(bank2, net.lin(p + ".attn1.to_v.weight"), trans_out=True, out=vbuf)
    ref.stale = False


def prepare_reference(net, cfg, refs, ehs, attn2_cache, attn2_slot=0):
    """Everything of a denoising forward that depends on the CLIP token and the reference banks only — the bank
    projections of the 16 hooked blocks and the collapsed attn2 vectors — refreshed in place, once per clip.  After it,
    a captured hipGraph of the forward (which reads those buffers) is valid for every DDIM step of the clip, the first
    one included."""
    for p in attention_paths(cfg):
        ref = refs.get(p)
        if ref is not None and ref.mode == "read" and ref.bank is not None and (ref.kref is None or ref.stale):
            project_reference_bank(net, p + ".transformer_blocks.0", ref, ref.bank.shape[-1] // cfg["attention_head_dim"])
    attn2_cache.get(net, cfg, ehs, refresh=True, slot=attn2_slot)


def transformer_block(net, p, h, Nf, T, heads, attn2_vec, rows_per_sample, ref=None, ref_index=None, stop_after_bank=False, dup=1):
    """(Temporal)BasicTransformerBlock under ReferenceAttentionControl
    (src/models/attention.py:383-445, src/models/mutual_self_attention.py:93-265).  h (Nf*T, C).
    dup = 2 (read mode, CFG pair with identical inputs — unet_forward's cfg_shared_input): h holds the Nf / 2 frames the two
    halves share; norm1 and the q / k / v projections run once, the attention reads them for both halves (frame n attends with
    the tokens of frame n % (Nf / 2), each with its own reference index), and from the attention output on the block carries all
    Nf frames: the output projection runs per half against the shared residual.  Returns (Nf*T, C)."""
    C = h.shape[1]
    d = C // heads
    M = h.shape[0]
    mode = ref.mode if ref is not None else "plain"
    assert dup == 1 or (dup == 2 and mode != "write" and M * 2 == Nf * T)
    qa = ops.attn_q_alpha(d)
    # Q token-major, multiplied by d^-1/2 log2(e) in its projection (the GEMM's alpha: still one fp16 rounding of the fp32
    # accumulator; the attention kernel then exponentiates q.k in base 2 with no per-score multiply); K HEAD-MAJOR (heads,
    # tokens, d): with a fused [to_q; to_k] projection every head's K row (2 d bytes) sits at a 4C-byte stride and the
    # attention kernel's K-tile reads touch 2-3x the cache lines (measured at 64x64, d = 40: 1.75 ms fused rows, 1.60 ms
    # separate matrices, 1.54 ms contiguous rows — against +12 us for the second GEMM launch); V transposed [C][Nf*T]
    wq, wk, wv = p + ".attn1.to_q.weight", p + ".attn1.to_k.weight", p + ".attn1.to_v.weight"
    if mode != "write" and heads == 8 and _FUSED_ROWS and ops.rowgemm320_supported(M, C):
        q, k, vt = ops.ln_qkv_projection(h, net.f32(p + ".norm1.weight"), net.f32(p + ".norm1.bias"), net.cat_lin((wq, wk, wv)),
                                         heads, qa)
    else:
        nh = ops.layernorm(h, net.f32(p + ".norm1.weight"), net.f32(p + ".norm1.bias"))
        if mode == "write":
            ref.written = nh.reshape(Nf, T, C)
            if stop_after_bank:
                return None
        q = ops.gemm(nh, net.lin(wq), alpha=qa)
        k = ops.gemm(nh, net.lin(wk), head_dim=d)
        vt = ops.gemm(nh, net.lin(wv), trans_out=True)  # V^T [C][Nf*T]
    kw = {}
    if mode == "read" and ref.bank is not None:
        if ref.kref is None or ref.stale:
            project_reference_bank(net, p, ref, d)
        assert ref.bank.shape[1] == T, "reference bank token count differs from the denoising latents"
        kw = dict(kref=ref.kref, ldkr=d, kref_head_stride=ref.kref.shape[0] * d, vtref=ref.vtref,
                  ldvtr=ref.vtref.shape[1], ref_index=ref_index[0], n_ref_frames=ref_index[1])
    a = ops.ref_attention(q, C, k, d, vt, vt.shape[1], Nf, T, heads, d, k_head_stride=M * d, q_log2_scaled=True,
                          frame_mod=Nf // dup if dup > 1 else 0, **kw)
    # attn1 out-proj + residual (+ the collapsed attn2: one vector per sample)
    if dup == 1:
        h = ops.gemm(a, net.lin(p + ".attn1.to_out.0.weight"), net.f32(p + ".attn1.to_out.0.bias"),
                     rowbias=attn2_vec, rows_per_group=rows_per_sample, residual=h)
    else:
        full = torch.empty((Nf * T, C), dtype=F16, device=h.device)
        for s_ in range(dup):           # one CFG half = one sample: its attn2 vector, the shared residual
            ops.gemm(a[s_ * M:(s_ + 1) * M], net.lin(p + ".attn1.to_out.0.weight"), net.f32(p + ".attn1.to_out.0.bias"),
                     rowbias=None if attn2_vec is None else attn2_vec[s_:s_ + 1], rows_per_group=rows_per_sample, residual=h,
                     out=full[s_ * M:(s_ + 1) * M])
        h = full
    return feed_forward(net, p + ".ff", h, p + ".norm3")


def spatial_transformer(net, p, x, heads, attn2_vec, frames_per_sample, ref=None, ref_index=None,
                        stop_after_bank=False, dup=1):
    """Transformer3DModel / Transformer2DModel (src/models/transformer_3d.py:103-169): x (N,H,W,C).
    dup = 2: x holds the N frames the two CFG halves share -> (2 N, H, W, C) (see transformer_block)."""
    N, H, W, C = x.shape
    T = H * W
    h = transformer_in(net, p, x)
    h = transformer_block(net, p + ".transformer_blocks.0", h, N * dup, T, heads, attn2_vec, frames_per_sample * T,
                          ref, ref_index, stop_after_bank, dup)
    if h is None:
        return None
    if dup == 1:
        out = ops.gemm(h, net.lin(p + ".proj_out.weight"), net.f32(p + ".proj_out.bias"), residual=x.reshape(N * T, C))
        return out.reshape(N, H, W, C)
    out = torch.empty((dup * N * T, C), dtype=F16, device=x.device)
    for s_ in range(dup):
        ops.gemm(h[s_ * N * T:(s_ + 1) * N * T], net.lin(p + ".proj_out.weight"), net.f32(p + ".proj_out.bias"),
                 residual=x.reshape(N * T, C), out=out[s_ * N * T:(s_ + 1) * N * T])
    return out.reshape(dup * N, H, W, C)


def motion_module(net, p, x, b, f, heads):
    """VanillaTemporalModule (src/models/motion_module.py:146-182,236-259,351-388): x (b*f,H,W,C)."""
    p = p + ".temporal_transformer"
    N, H, W, C = x.shape
    T = H * W
    d = C // heads
    h = transformer_in(net, p, x)
    bp = p + ".transformer_blocks.0"
    i = 0
    while net.has(bp + f".attention_blocks.{i}.to_q.weight"):
        ap = bp + f".attention_blocks.{i}"
        pe_name = ap + ".pos_encoder.pe"
        pe = net.pe(pe_name, C) if net.has(pe_name) else None
        if pe is not None and f > pe.shape[0]:
            raise ValueError(f"video_length {f} exceeds temporal_position_encoding_max_len {pe.shape[0]}")
        # LN + positional encoding of the frame (added to the attention INPUT: q, k and v see it; the
        # residual below uses the un-encoded h — src/models/motion_module.py:244-254,365-366)
        wn = (ap + ".to_q.weight", ap + ".to_k.weight", ap + ".to_v.weight")
        if _FUSED_TEMPORAL and ops.temporal_qkv_attention_supported(f, T, C, heads):
            a = ops.temporal_qkv_attention(h, net.f32(bp + f".norms.{i}.weight"),
                                           net.beta_pe(bp + f".norms.{i}", pe_name if pe is not None else None, C, f),
                                           net.tqkv(wn), b, f, T, heads)
            h = ops.gemm(a, net.lin(ap + ".to_out.0.weight"), net.f32(ap + ".to_out.0.bias"), residual=h)
            i += 1
            continue
        nh = ops.layernorm(h, net.f32(bp + f".norms.{i}.weight"), net.f32(bp + f".norms.{i}.bias"), pe=pe,
                           rows_per_frame=T, frames=f)
        qkv = ops.gemm(nh, net.cat_lin(wn))
        a = ops.temporal_attention(qkv, b, f, T, heads, d)
        h = ops.gemm(a, net.lin(ap + ".to_out.0.weight"), net.f32(ap + ".to_out.0.bias"), residual=h)
        i += 1
    h = feed_forward(net, bp + ".ff", h, bp + ".ff_norm")
    out = ops.gemm(h, net.lin(p + ".proj_out.weight"), net.f32(p + ".proj_out.bias"), residual=x.reshape(N * T, C))
    return out.reshape(N, H, W, C)


# ----------------------------------------------------------------------------------------------------
# UNets
# ----------------------------------------------------------------------------------------------------

def timestep_sinusoid(t, batch, dim, device, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    """diffusers Timesteps / get_timestep_embedding (src/models/unet_3d.py:95-98,463-466): tiny host
    arithmetic, fp32."""
    if torch.is_tensor(t):
        tt = t.detach().reshape(-1).to("cpu", torch.float32)
    else:
        tt = torch.tensor([float(t)], dtype=torch.float32)
    tt = tt.expand(batch) if tt.numel() == 1 else tt
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = tt[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb.to(device, non_blocking=True)


def resnet_names(cfg):
    nblk = len(cfg["block_out_channels"])
    lpb = cfg["layers_per_block"]
    names = [f"down_blocks.{i}.resnets.{j}" for i in range(nblk) for j in range(lpb)]
    names += ["mid_block.resnets.0", "mid_block.resnets.1"]
    names += [f"up_blocks.{i}.resnets.{j}" for i in range(nblk) for j in range(lpb + 1)]
    return names


def attention_paths(cfg):
    """spatial transformer paths in execution order"""
    nblk = len(cfg["block_out_channels"])
    lpb = cfg["layers_per_block"]
    out = [f"down_blocks.{i}.attentions.{j}" for i in range(nblk) if _DOWN_HAS_ATTN[i] for j in range(lpb)]
    out += ["mid_block.attentions.0"]
    out += [f"up_blocks.{i}.attentions.{j}" for i in range(nblk) if _UP_HAS_ATTN[i] for j in range(lpb + 1)]
    return out


class Attn2Cache:
    """to_out(to_v(e)) of every spatial transformer block for a CLIP embedding e: constant over DDIM steps.

    The vectors live in one buffer set per embedding shape, allocated once and rewritten IN PLACE (captured hipGraphs
    of the forward read them).  There is no content cache: `get(..., refresh=True)` — the default of every forward —
    recomputes them from the tensor it is given (32 tiny launches); the pipeline's graph runner passes refresh=False
    for the steps after the clip's first one, whose eager forward has just written them.  (A pointer / version key,
    as in round 1, does not identify a tensor's content once the allocator recycles the block.)"""

    def __init__(self):
        self.pool = {}   # (b, device) -> {path: fp32 (b, C)}

    def drop(self):
        self.pool = {}

    def get(self, net, cfg, ehs, refresh=True, slot=0):
        """slot: distinct buffer sets for same-sized embeddings that are live at the same time (the two CFG halves of a
        step running as separate forwards on two streams)"""
        key = (int(ehs.shape[0]), str(net.device), int(slot))
        vecs = self.pool.get(key)
        if vecs is not None and not refresh:
            return vecs
        e = ehs.detach().reshape(ehs.shape[0], -1).to(net.device, F32).contiguous()  # (b, D): sequence length 1
        old = vecs or {}
        vecs = {}
        for p in attention_paths(cfg):
            a = p + ".transformer_blocks.0.attn2"
            v = ops.linear_small(e, net.lin(a + ".to_v.weight"))
            vecs[p] = ops.linear_small(v, net.lin(a + ".to_out.0.weight"), net.f32(a + ".to_out.0.bias"),
                                       out=old.get(p))
        self.pool[key] = vecs
        return vecs


def unet_forward(net, cfg, x, b, f, t, ehs, attn2_cache, refs, with_motion, ref_index=None, pose_nhwc=None,
                 final=True, stop_after_last_bank=False, temb_in=None, attn2_refresh=True, tap=None, attn2_slot=0,
                 cfg_shared_input=False):
    """UNet3DConditionModel.forward (src/models/unet_3d.py:399-580) / the ReferenceNet
    UNet2DConditionModel.forward (src/models/unet_2d_condition.py:872-1308, f = 1, no motion modules).

    x (b*f, h, w, 4) fp16 channels-last.  ehs (b, 1, D).  refs: {path: RefState}.  ref_index: (int32 tensor (b*f,), n) —
    reference sample per frame, -1 for the CFG-unconditional frames that attend to self only
    (src/models/mutual_self_attention.py:77-85,166-186), and the number n of frames that do have one.  pose_nhwc: list of 5 channels-last tensors or None.  Returns (b*f, h, w, out_channels) fp16 (or the last hidden state if
    `final` is False; None if `stop_after_last_bank`).  tap(name, x): optional observer of every block output
    (channels-last; tools/bisect_parity.py compares them with the oracle's, block by block).
    cfg_shared_input: the caller GUARANTEES that the b = 2 samples are a classifier-free-guidance pair built by duplication —
    identical x, pose features and timestep (pipeline_pose2vid_long.py:521-536: `latent_model_input = torch.cat([latents] * 2)`,
    the pose features repeated, one t); they differ in encoder_hidden_states (which enters behind attn1 only) and in their
    reference index.  Then everything in front of the first reference attention — the first ResnetBlock3D, the transformer's
    GroupNorm + proj_in, norm1 and the q / k / v projections — is the same for both halves and runs ONCE on f frames.
    """
    if ehs.shape[1] != 1:
        raise NotImplementedError("encoder_hidden_states with sequence length != 1: the pose2vid path feeds "
                                  "one CLIP image token (pipeline_pose2vid_long.py:385)")
    boc = tuple(cfg["block_out_channels"])
    nblk, lpb = len(boc), cfg["layers_per_block"]
    heads = cfg["attention_head_dim"]
    eps = cfg["norm_eps"]
    groups = cfg["norm_num_groups"]
    N, H, W, _ = x.shape
    assert N == b * f
    # ResnetBlock3D norms and conv_norm_out: per frame (InflatedGroupNorm) or over the sample's f frames (nn.GroupNorm on
    # (b, c, f, h, w): use_inflated_groupnorm=False, configs/inference/inference_v1.yaml); the Transformer3D / motion-module
    # norms see (b f) c h w in both variants (src/models/transformer_3d.py:115-124, src/models/motion_module.py:151-156)
    gn_frames = 1 if (not with_motion or cfg.get("use_inflated_groupnorm", True)) else f

    # temb_in: the sinusoid already on the device (fp32 (b, C0)) — lets a captured hipGraph of this forward be
    # replayed for every DDIM step by rewriting that one buffer
    emb = temb_in if temb_in is not None else timestep_sinusoid(t, b, boc[0], net.device,
                                                               cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0))
    emb = ops.linear_small(emb, net.lin("time_embedding.linear_1.weight"), net.f32("time_embedding.linear_1.bias"))
    emb = ops.linear_small(emb, net.lin("time_embedding.linear_2.weight"), net.f32("time_embedding.linear_2.bias"),
                           silu_in=True)
    Wt, bt, toffs = net.temb_stack([n + ".time_emb_proj" for n in resnet_names(cfg)])
    temb_all = ops.linear_small(emb, Wt, bt, silu_in=True)  # (b, sum Cout) fp32

    a2 = attn2_cache.get(net, cfg, ehs, attn2_refresh, attn2_slot)
    last_path = attention_paths(cfg)[-1]

    def see(p, x):
        if tap is not None and x is not None:
            tap(p, x)
        return x

    def res(p, x, skip=None):
        o, c = toffs[p + ".time_emb_proj"]
        hw = x.shape[1] * x.shape[2]
        return see(p, resnet(net, p, x, skip, temb_all[:, o:o + c], f * hw, eps, groups, gn_frames))

    def attn(p, x):
        return see(p, spatial_transformer(net, p, x, heads, a2[p], f, refs.get(p), ref_index,
                                          stop_after_bank=stop_after_last_bank and p == last_path))

    def mm(p, x):
        if with_motion and net.has(p + ".temporal_transformer.proj_in.weight"):
            return see(p, motion_module(net, p, x, b, f, cfg["motion_module_kwargs"]["num_attention_heads"]))
        return x

    def add_pose(x, i):
        if pose_nhwc is None:
            return x
        return ops.add(x, pose_nhwc[i])

    x = ops.conv_direct(x, net.conv_direct("conv_in.weight"), net.f32("conv_in.bias"), boc[0], 3, 1, 1,
                        residual=None if pose_nhwc is None else pose_nhwc[0])
    see("conv_in", x)
    skips = [x]
    p00 = "down_blocks.0.attentions.0"
    share = (cfg_shared_input and _SHARE_CFG_PREFIX and b == 2 and _DOWN_HAS_ATTN[0] and gn_frames == 1 and tap is None and
             not stop_after_last_bank and refs.get(p00) is not None and refs[p00].mode == "read" and ref_index is not None)
    if share and _CHECK_CFG_PREFIX and not torch.cuda.is_current_stream_capturing():
        # ANIP_CHECK_CFG_PREFIX=1 (debug): the caller's guarantee, verified — a host sync per call, never on by default
        same = bool(torch.equal(x[:f], x[f:])) and bool(torch.equal(temb_all[0], temb_all[1]))
        if not same:
            raise ValueError("unet_forward(cfg_shared_input=True): the two CFG halves differ in x or in the time embedding")
    for i in range(nblk):
        for j in range(lpb):
            if share and i == 0 and j == 0:
                # the CFG halves' shared prefix: first resnet + norm / proj_in / norm1 / q | k | v on the f frames of ONE half
                o, c = toffs["down_blocks.0.resnets.0.time_emb_proj"]
                r0 = resnet(net, "down_blocks.0.resnets.0", x[:f], None, temb_all[:1, o:o + c], f * H * W, eps, groups, 1)
                x = spatial_transformer(net, p00, r0, heads, a2[p00], f, refs.get(p00), ref_index, dup=2)
                x = mm("down_blocks.0.motion_modules.0", x)
                skips.append(x)
                continue
            x = res(f"down_blocks.{i}.resnets.{j}", x)
            if _DOWN_HAS_ATTN[i]:
                x = attn(f"down_blocks.{i}.attentions.{j}", x)
            x = mm(f"down_blocks.{i}.motion_modules.{j}", x)
            skips.append(x)
        if i != nblk - 1:
            pn = f"down_blocks.{i}.downsamplers.0.conv"
            x = see(pn[:-5], ops.conv3x3(x, net.conv3(pn + ".weight"), net.f32(pn + ".bias"), stride=2, pad=1))
            skips.append(x)
        x = add_pose(x, i + 1)
    x = res("mid_block.resnets.0", x)
    x = attn("mid_block.attentions.0", x)
    x = mm("mid_block.motion_modules.0", x)
    x = res("mid_block.resnets.1", x)
    for i in range(nblk):
        for j in range(lpb + 1):
            x = res(f"up_blocks.{i}.resnets.{j}", x, skips.pop())
            if _UP_HAS_ATTN[i]:
                x = attn(f"up_blocks.{i}.attentions.{j}", x)
                if x is None:
                    return None
            x = mm(f"up_blocks.{i}.motion_modules.{j}", x)
        if i != nblk - 1:
            pn = f"up_blocks.{i}.upsamplers.0.conv"
            x = see(pn[:-5], ops.conv3x3(x, net.conv3(pn + ".weight"), net.f32(pn + ".bias"), upsample=True))
    if not final or not net.has("conv_out.weight"):
        return x
    C = x.shape[-1]
    h = ops.groupnorm(x.reshape(N, H * W, C), net.f32("conv_norm_out.weight"), net.f32("conv_norm_out.bias"), groups,
                      eps, True, frames_per_stat=gn_frames)
    return see("conv_out", ops.conv3x3(h.reshape(N, H, W, C), net.conv3("conv_out.weight"), net.f32("conv_out.bias")))


# ----------------------------------------------------------------------------------------------------
# VAE (diffusers AutoencoderKL, sd-vae-ft-mse topology)
# ----------------------------------------------------------------------------------------------------

def _vae_mid(net, p, x, see=None):
    see = see or (lambda name, x: x)
    x = see(p + ".resnets.0", resnet(net, p + ".resnets.0", x, None, None, 0, 1e-6))
    N, H, W, C = x.shape
    T = H * W
    a = p + ".attentions.0"
    t = ops.groupnorm(x.reshape(N, T, C), net.f32(a + ".group_norm.weight"), net.f32(a + ".group_norm.bias"), 32,
                      1e-6, False)
    t2 = t.reshape(N * T, C)
    qk = ops.gemm(t2, net.cat_lin((a + ".to_q.weight", a + ".to_k.weight")),
                  torch.cat([net.f32(a + ".to_q.bias"), net.f32(a + ".to_k.bias")]))
    qk3 = qk.reshape(N, T, 2 * C)
    # single head, d = C (512): scores via batched GEMM, fp32 softmax (upcast_softmax), P V via batched GEMM
    s = ops.gemm(qk3[:, :, :C], qk3[:, :, C:], None, batch=N, out_f32=True, alpha=C ** -0.5)
    pm = ops.softmax_rows(s)
    vt = ops.gemm(net.lin(a + ".to_v.weight"), t, None, batch=N)  # (N, C, T) = W_v t^T per frame
    o = ops.gemm(pm, vt, None, batch=N)  # (N, T, C)
    x = see(a, ops.gemm(o.reshape(N * T, C), net.lin(a + ".to_out.0.weight"), net.vae_attn_out(a),
                        residual=x.reshape(N * T, C)).reshape(N, H, W, C))
    return see(p + ".resnets.1", resnet(net, p + ".resnets.1", x, None, None, 0, 1e-6))


def vae_decode(net, cfg, z, tap=None):
    """AutoencoderKL.decode(z).sample, batched over frames (src/pipelines/pipeline_pose2vid_long.py:119-120
    decodes frame by frame; frames are independent).  z (N, h, w, 4) fp16 -> (N, 8h, 8w, 3) fp16.
    tap(name, x): optional observer of every block output (tests/bisect_parity.py)."""
    def see(name, x):
        if tap is not None:
            tap(name, x)
        return x

    nb = len(cfg["block_out_channels"])
    x = ops.conv_direct(z, net.conv_direct("post_quant_conv.weight"), net.f32("post_quant_conv.bias"),
                        cfg["latent_channels"], 1, 1, 0)
    x = see("decoder.conv_in", ops.conv_direct(x, net.conv_direct("decoder.conv_in.weight"), net.f32("decoder.conv_in.bias"),
                                               cfg["block_out_channels"][-1], 3, 1, 1))
    x = _vae_mid(net, "decoder.mid_block", x, see)
    for i in range(nb):
        for j in range(cfg["layers_per_block"] + 1):
            x = see(f"decoder.up_blocks.{i}.resnets.{j}", resnet(net, f"decoder.up_blocks.{i}.resnets.{j}", x, None, None, 0, 1e-6))
        if i != nb - 1:
            pn = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            x = see(pn[:-5], ops.conv3x3(x, net.conv3(pn + ".weight"), net.f32(pn + ".bias"), upsample=True))
    N, H, W, C = x.shape
    h = see("decoder.conv_norm_out", ops.groupnorm(x.reshape(N, H * W, C), net.f32("decoder.conv_norm_out.weight"),
                                                   net.f32("decoder.conv_norm_out.bias"), 32, 1e-6, True).reshape(N, H, W, C))
    return see("decoder.conv_out", ops.conv3x3(h, net.conv3("decoder.conv_out.weight"), net.f32("decoder.conv_out.bias")))


def vae_encode_mean(net, cfg, x):
    """AutoencoderKL.encode(x).latent_dist.mean (src/pipelines/pipeline_pose2vid_long.py:430).
    x (N, H, W, 3) fp16 -> (N, H/8, W/8, latent_channels) fp16."""
    nb = len(cfg["block_out_channels"])
    x = ops.conv_direct(x, net.conv_direct("encoder.conv_in.weight"), net.f32("encoder.conv_in.bias"),
                        cfg["block_out_channels"][0], 3, 1, 1)
    for i in range(nb):
        for j in range(cfg["layers_per_block"]):
            x = resnet(net, f"encoder.down_blocks.{i}.resnets.{j}", x, None, None, 0, 1e-6)
        if i != nb - 1:
            pn = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            x = ops.conv3x3(x, net.conv3(pn + ".weight"), net.f32(pn + ".bias"), stride=2, pad=0, pad_hi=1)
    x = _vae_mid(net, "encoder.mid_block", x)
    N, H, W, C = x.shape
    h = ops.groupnorm(x.reshape(N, H * W, C), net.f32("encoder.conv_norm_out.weight"),
                      net.f32("encoder.conv_norm_out.bias"), 32, 1e-6, True)
    m = ops.conv3x3(h.reshape(N, H, W, C), net.conv3("encoder.conv_out.weight"), net.f32("encoder.conv_out.bias"))
    lat2 = m.shape[-1]
    q = ops.gemm(m.reshape(N * H * W, lat2), net.lin("quant_conv.weight"), net.f32("quant_conv.bias"))
    return q.reshape(N, H, W, lat2)[..., : cfg["latent_channels"]].contiguous()


# ----------------------------------------------------------------------------------------------------
# PoseGuider (src/models/pose_guider.py:13-162)
# ----------------------------------------------------------------------------------------------------

def _pose_conv_bn_relu(net, name, k, x, co, ks, stride, pad, training):
    """Conv2d -> BatchNorm2d -> ReLU (entries 3k, 3k+1, 3k+2 of an nn.Sequential, pose_guider.py:19-85)."""
    N, H, W, ci = x.shape
    wn, bn_ = f"{name}.{3 * k}.weight", f"{name}.{3 * k}.bias"
    if ks == 3 and ci % 64 == 0:
        y = ops.conv3x3(x, net.conv3(wn), net.f32(bn_), stride=stride, pad=pad)       # MFMA implicit GEMM
    else:
        y = ops.conv_direct(x, net.conv_direct(wn), net.f32(bn_), co, ks, stride, pad)  # small-channel stem
    n = f"{name}.{3 * k + 1}"
    rm = rv = None
    if not training:
        rm, rv = net.f32(n + ".running_mean"), net.f32(n + ".running_var")
    Nn, Ho, Wo, _ = y.shape
    return ops.batchnorm(y.reshape(Nn * Ho * Wo, co), net.f32(n + ".weight"), net.f32(n + ".bias"), rm, rv, 1e-5,
                         True).reshape(Nn, Ho, Wo, co)


def _pose_self_attn(net, p, x, heads=16):
    """pose_guider.Transformer2DModel (pose_guider.py:165-308) with cross_attention_dim=None: GroupNorm(1e-6) ->
    1x1 proj_in -> [LN, 16-head self-attention (d = 88), +res, LN, GEGLU FF, +res] -> 1x1 proj_out -> +residual.
    Its second argument (ref_x) never reaches the arithmetic (no attn2), so it is not evaluated."""
    N, H, W, C = x.shape
    T = H * W
    h = transformer_in(net, p, x)
    h = transformer_block(net, p + ".transformer_blocks.0", h, N, T, heads, None, 0)
    out = ops.gemm(h, net.lin(p + ".proj_out.weight"), net.f32(p + ".proj_out.bias"), residual=x.reshape(N * T, C))
    return out.reshape(N, H, W, C)


def pose_guider_forward(net, stacks, x, training, use_ca):
    """PoseGuider.forward (src/models/pose_guider.py:124-162) on channels-last frames.
    x (N, H, W, 3) fp16 -> 5 feature maps (N, h, w, C) at 1/8, 1/16, 1/32, 1/64, 1/64 resolution."""
    def stack(name, x):
        _cin, layers = stacks[name]
        for k, (co, ks, stride, pad) in enumerate(layers):
            x = _pose_conv_bn_relu(net, name, k, x, co, ks, stride, pad, training)
        return x

    x = stack("conv_layers", x)
    N, H, W, C = x.shape
    scale = net.scalar("scale")
    # final_proj(x) * scale  ==  scale * (x W^T) + scale * b   (pose_guider.py:129-131)
    x = ops.gemm(x.reshape(N * H * W, C), net.lin("final_proj.weight"), net.scaled_f32("final_proj.bias", scale),
                 alpha=scale).reshape(N, H, W, -1)
    fea = [x]
    for i in range(1, 5):
        x = stack(f"conv_layers_{i}", x)
        if use_ca:
            x = _pose_self_attn(net, f"cross_attn{i}", x)
        fea.append(x)
    return fea


# ----------------------------------------------------------------------------------------------------
# CLIP vision tower (the reference image's embedding, once per clip)
# ----------------------------------------------------------------------------------------------------

def clip_patch_weight(net, Kp):
    """patch_embedding Conv2d(3, C, k = s = patch, bias=False) as a GEMM matrix fp16 [C][Kp]: (c, ky, kx)-ordered columns,
    zero-padded to the Kp columns of `clip_patches`"""
    k = ("clippatch", int(Kp))
    if k not in net.t:
        w = net._raw("vision_model.embeddings.patch_embedding.weight")
        w = w.reshape(w.shape[0], -1).to(net.device, F16)
        wp = torch.zeros((w.shape[0], Kp), dtype=F16, device=net.device)
        wp[:, :w.shape[1]] = w
        net.t[k] = wp
    return net.t[k]


def clip_token_table(net):
    """fp16 [1 + P][C]: row 0 = class_embedding + position_embedding[0], row 1 + i = position_embedding[1 + i] — the residual of
    the patch GEMM (the class token's GEMM row is all zeros)"""
    k = ("cliptok",)
    if k not in net.t:
        pos = net._raw("vision_model.embeddings.position_embedding.weight").to(net.device, F32).clone()
        pos[0] += net._raw("vision_model.embeddings.class_embedding").to(net.device, F32)
        net.t[k] = pos.to(F16).contiguous()
    return net.t[k]


def clip_vision_forward(net, cfg, patches):
    """`CLIPVisionModelWithProjection(pixel_values).image_embeds` of transformers (the call of
    src/pipelines/pipeline_pose2vid_long.py:379-385) on the HIP kernels: ViT with a class token, pre-LayerNorm, L x [LN ->
    q | k | v (+ bias) -> softmax attention over the 1 + P tokens -> out_proj + residual -> LN -> fc1 + quick-GELU -> fc2 +
    residual], post-LayerNorm of the class token, bias-free visual projection.

    patches (B * (1 + P), Kp) fp16: per image one all-zero row (the class token's slot) followed by its P im2col'd patches in
    row-major patch order, each (c, ky, kx)-ordered and zero-padded to Kp (`clip_vision.patch_rows`, built on the host from the
    `CLIPImageProcessor` output, which lives there anyway).  cfg: hidden_size, num_attention_heads, num_hidden_layers,
    layer_norm_eps.  Returns (B, projection_dim) fp16."""
    C, heads, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    d = C // heads
    tok = clip_token_table(net)
    T = tok.shape[0]
    M = patches.shape[0]
    B = M // T
    assert B * T == M, f"patch rows {M} are not a multiple of the {T} tokens per image"
    wpe = clip_patch_weight(net, patches.shape[1])
    if B == 1:
        h = ops.gemm(patches, wpe, residual=tok)
    else:
        h = torch.empty((M, C), dtype=F16, device=patches.device)
        for b_ in range(B):
            ops.gemm(patches[b_ * T:(b_ + 1) * T], wpe, residual=tok, out=h[b_ * T:(b_ + 1) * T])
    vm = "vision_model."
    h = ops.layernorm(h, net.f32(vm + "pre_layrnorm.weight"), net.f32(vm + "pre_layrnorm.bias"), eps)
    qa = ops.attn_q_alpha(d)
    Mp = (M + 7) // 8 * 8           # V^T row pitch: 16-B aligned rows; ONE buffer for all layers, its pad columns stay zero
    vt = torch.zeros((C, Mp), dtype=F16, device=patches.device)
    for i in range(cfg["num_hidden_layers"]):
        p = f"{vm}encoder.layers.{i}."
        a = p + "self_attn."
        nh = ops.layernorm(h, net.f32(p + "layer_norm1.weight"), net.f32(p + "layer_norm1.bias"), eps)
        # q carries d^-1/2 log2(e) (weights through the GEMM's alpha, the bias pre-multiplied): the attention kernel
        # exponentiates in base 2; k head-major, v transposed — the layouts of the UNets' reference attention
        q = ops.gemm(nh, net.lin(a + "q_proj.weight"), net.scaled_f32(a + "q_proj.bias", qa), alpha=qa)
        k = ops.gemm(nh, net.lin(a + "k_proj.weight"), net.f32(a + "k_proj.bias"), head_dim=d)
        ops.gemm(nh, net.lin(a + "v_proj.weight"), net.f32(a + "v_proj.bias"), trans_out=True, out=vt, ldo=Mp)
        o = ops.ref_attention(q, C, k, d, vt, Mp, B, T, heads, d, k_head_stride=M * d, q_log2_scaled=True)
        h = ops.gemm(o, net.lin(a + "out_proj.weight"), net.f32(a + "out_proj.bias"), residual=h)
        nh = ops.layernorm(h, net.f32(p + "layer_norm2.weight"), net.f32(p + "layer_norm2.bias"), eps)
        g = ops.gemm(nh, net.lin(p + "mlp.fc1.weight"), net.f32(p + "mlp.fc1.bias"), act=2)
        h = ops.gemm(g, net.lin(p + "mlp.fc2.weight"), net.f32(p + "mlp.fc2.bias"), residual=h)
    pooled = h[:1] if B == 1 else h.reshape(B, T, C)[:, 0].contiguous()
    pooled = ops.layernorm(pooled, net.f32(vm + "post_layernorm.weight"), net.f32(vm + "post_layernorm.bias"), eps)
    return ops.gemm(pooled, net.lin("visual_projection.weight"))
