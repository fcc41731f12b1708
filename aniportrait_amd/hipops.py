"""Tensor-level wrappers over the C ABI (include/aniportrait_hip.h).

PyTorch is used for device memory and streams only: every function here takes CUDA(HIP) tensors,
passes raw pointers + sizes to libaniportrait_hip.so on torch's current stream and returns torch
tensors that own the outputs.  Activations are channels-last fp16: (N, H, W, C) == (N*H*W, C).
No fallbacks: a missing library or a non-GPU tensor raises.
"""
import ctypes as C
import os

import torch

from . import _lib as L

F16 = torch.float16
F32 = torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _req(t, dtype, name):
    if not t.is_cuda:
        raise L.HipLibraryError(f"{name}: expected a GPU tensor (the hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


# ------------------------------------------------------------------------------------------------
# per-kernel profiling (HIP events in the library + algorithmic work counted here)
# ------------------------------------------------------------------------------------------------
K_GEMM, K_CONV3X3, K_GN_STATS, K_GN_APPLY, K_LAYERNORM, K_REF_ATTN, K_TEMPORAL_ATTN, K_SOFTMAX = range(8)
K_CONV_SMALL, K_LINEAR_SMALL, K_ELEMENTWISE, K_BATCHNORM = 8, 9, 10, 11
_FLOP_KERNELS = (K_GEMM, K_CONV3X3, K_REF_ATTN)
_WORK = None  # kernel id -> algorithmic work (flops for 0,1,5; bytes otherwise) while profiling
_CALLS = None  # [(kernel id, shape descriptor, work)] in launch order while profiling
_ABYTES = None  # parallel to _CALLS: algorithmic HBM bytes (operands read once + outputs written once) per call


def _work(kid, amount, desc="", abytes=None):
    """one call per library-side event bracket, in launch order (profile() pairs them up by position).  `abytes`:
    algorithmic bytes of a FLOP-counted kernel (byte-counted kernels: `amount` itself)"""
    if _WORK is not None:
        _WORK[kid] = _WORK.get(kid, 0) + amount
        _CALLS.append((kid, desc, amount))
        if _ABYTES is not None:
            _ABYTES.append(int(abytes) if abytes is not None else (0 if kid in _FLOP_KERNELS else int(amount)))


class trace_calls:
    """`with trace_calls() as t: ...; t.calls` -> [{family, shape, work, unit, algorithmic_bytes}] of every wrapper call
    in launch order, WITHOUT the library-side event brackets (for rocprofv3 --pmc passes: tools/pmc_unet_step.py writes
    this list next to the counter CSVs and tools/pmc_summarize.py pairs it with the dispatch rows by order, which is
    what tells a 3x3 conv from a Linear and one shape from another — the kernel symbol alone cannot)."""

    def __enter__(self):
        global _WORK, _CALLS, _ABYTES
        _WORK, _CALLS, _ABYTES = {}, [], []
        self.calls = None
        return self

    def __exit__(self, *exc):
        global _WORK, _CALLS, _ABYTES
        lib = L.load()
        self.calls = [dict(family=lib.anip_profile_kernel_name(k).decode(), shape=desc, work=w,
                           unit="FLOP" if k in _FLOP_KERNELS else "B", algorithmic_bytes=ab)
                      for (k, desc, w), ab in zip(_CALLS, _ABYTES)]
        _WORK = _CALLS = _ABYTES = None
        return False


class profile:
    """`with profile() as p: ...; p.result` -> {kernel name: {launches, ms, work, unit, rate}}.
    Every kernel launch inside the block is bracketed by HIP events on its stream (library side);
    `work` is the algorithmic FLOP (contractions) or byte (HBM-bound kernels) count of those launches.
    `p.by_shape`: the same, keyed by (kernel name, shape descriptor), when the per-launch records line up
    with the wrapper calls."""

    def __enter__(self):
        global _WORK, _CALLS
        lib = L.load()
        torch.cuda.synchronize()
        lib.anip_profile_collect(0, None, None)  # drop stale records
        lib.anip_profile_enable(1)
        _WORK, _CALLS = {}, []
        self.result = None
        self.by_shape = None
        return self

    def __exit__(self, *exc):
        global _WORK, _CALLS
        lib = L.load()
        lib.anip_profile_enable(0)
        n = L.N_KERNEL_IDS
        launches = (C.c_int64 * n)()
        ms = (C.c_double * n)()
        work, calls = _WORK, _CALLS
        _WORK = _CALLS = None
        cap = len(calls) + 16
        rk = (C.c_int * cap)()
        rms = (C.c_float * cap)()
        nrec = C.c_int64(0)
        rc = lib.anip_profile_collect_records(n, launches, ms, cap, rk, rms, C.byref(nrec))
        if exc[0] is None:
            L.check(rc, "anip_profile_collect_records")
        res = {}
        for k in range(n):
            if launches[k] == 0:
                continue
            flops = k in _FLOP_KERNELS
            w = work.get(k, 0)
            t = ms[k] * 1e-3
            res[lib.anip_profile_kernel_name(k).decode()] = dict(
                id=k, launches=int(launches[k]), ms=ms[k], work=w, unit="TFLOP/s" if flops else "GB/s",
                rate=(w / t / (1e12 if flops else 1e9)) if t > 0 else 0.0)
        self.result = res
        self.records = None   # [(kernel id, shape descriptor, work, ms)] in launch order
        if nrec.value == len(calls) and all(rk[i] == calls[i][0] for i in range(len(calls))):
            self.records = [(k, desc, w, float(rms[i])) for i, (k, desc, w) in enumerate(calls)]
            agg = {}
            for i, (k, desc, w) in enumerate(calls):
                a = agg.setdefault((k, desc), [0, 0.0, 0])
                a[0] += 1
                a[1] += float(rms[i])
                a[2] += w
            self.by_shape = [
                dict(kernel=lib.anip_profile_kernel_name(k).decode(), shape=desc, launches=c, ms=t, work=w,
                     unit="TFLOP/s" if k in _FLOP_KERNELS else "GB/s",
                     rate=(w / (t * 1e-3) / (1e12 if k in _FLOP_KERNELS else 1e9)) if t > 0 else 0.0)
                for (k, desc), (c, t, w) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
        return False


class _MergedProfile:
    pass


def merge_profiles_min(profs):
    """Combine profiles of IDENTICAL launch sequences by taking, per launch, the minimum of its event times: `result`
    and `by_shape` as in `profile`.  Falls back to the first profile when the records do not line up."""
    lib = L.load()
    first = profs[0]
    if any(p.records is None for p in profs) or any(
            len(p.records) != len(first.records) or any(a[:2] != b[:2] for a, b in zip(p.records, first.records)) for p in profs[1:]):
        return first
    out = _MergedProfile()
    recs = [(k, desc, w, min(p.records[i][3] for p in profs)) for i, (k, desc, w, _) in enumerate(first.records)]
    res, agg = {}, {}
    for k, desc, w, ms in recs:
        name = lib.anip_profile_kernel_name(k).decode()
        flops = k in _FLOP_KERNELS
        r = res.setdefault(name, dict(id=k, launches=0, ms=0.0, work=0, unit="TFLOP/s" if flops else "GB/s", rate=0.0))
        r["launches"] += 1
        r["ms"] += ms
        r["work"] += w
        a = agg.setdefault((k, desc), [0, 0.0, 0])
        a[0] += 1
        a[1] += ms
        a[2] += w
    for r in res.values():
        t = r["ms"] * 1e-3
        r["rate"] = (r["work"] / t / (1e12 if r["unit"] == "TFLOP/s" else 1e9)) if t > 0 else 0.0
    out.result, out.records = res, recs
    out.by_shape = [
        dict(kernel=lib.anip_profile_kernel_name(k).decode(), shape=desc, launches=c, ms=t, work=w,
             unit="TFLOP/s" if k in _FLOP_KERNELS else "GB/s",
             rate=(w / (t * 1e-3) / (1e12 if k in _FLOP_KERNELS else 1e9)) if t > 0 else 0.0)
        for (k, desc), (c, t, w) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    return out


def device_info():
    lib = L.load()
    buf = C.create_string_buffer(64)
    ncu = C.c_int(0)
    L.check(lib.anip_device_info(buf, 64, C.byref(ncu)), "anip_device_info")
    return buf.value.decode(), ncu.value


# ------------------------------------------------------------------------------------------------
# weight packing (host side, once)
# ------------------------------------------------------------------------------------------------

# K order of the implicit-GEMM 3x3 convolution (anip_gemm_params.conv):
#   1  tap-major          W [Cout][ky][kx][Cin]: all channels of tap 0, then tap 1, ...
#   2  channel-block-major W [Cout][Cin/64][ky][kx][64]: the nine taps of a 64-channel block are CONSECUTIVE K-tiles, so
#      the nine shifted re-reads of an input line follow each other within ~9 K-tiles instead of Cin/64 x 9 apart — the
#      re-reads then hit the XCD's 4 MB L2 (32 CUs x 245 KB of live input rows + the weights do not fit it in tap-major
#      order at 64x64 x 320 channels; rocprofv3 FETCH_SIZE showed 4x the algorithmic bytes).  Needs Cin % 64 == 0.
CONV_KORDER = int(os.environ.get("ANIP_CONV_KORDER", "2"))


def conv_korder(cin):
    return 2 if (CONV_KORDER == 2 and cin % 64 == 0) else 1


def pack_conv3x3(w, order=None):
    """torch conv weight [Cout, Cin, 3, 3] -> [Cout, 9*Cin] in the K order `order` (default: conv_korder(Cin))."""
    co, ci, kh, kw = w.shape
    order = conv_korder(ci) if order is None else order
    if order == 2:
        return w.reshape(co, ci // 64, 64, kh, kw).permute(0, 1, 3, 4, 2).reshape(co, kh * kw * ci).contiguous()
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def unpack_conv3x3(wp, cin, order=None):
    """inverse of pack_conv3x3: [Cout, 9*Cin] -> [Cout, Cin, 3, 3] (tests / the CPU emulation of the wrappers)"""
    co = wp.shape[0]
    order = conv_korder(cin) if order is None else order
    if order == 2:
        return wp.reshape(co, cin // 64, 3, 3, 64).permute(0, 1, 4, 2, 3).reshape(co, cin, 3, 3)
    return wp.reshape(co, 3, 3, cin).permute(0, 3, 1, 2)


def pack_geglu(w, b):
    """FeedForward GEGLU projection [8C, C] (+bias [8C]) -> rows grouped per 16 output columns as
    [16 x value | 16 x gate]: the value and gate MFMA tiles of one output column then sit in the same lane of
    the GEMM epilogue, which applies value * gelu(gate) in registers."""
    n2, k = w.shape
    n = n2 // 2
    assert n % 64 == 0, "GEGLU inner dim must be a multiple of 64"
    wv, wg = w[:n].reshape(n // 16, 16, k), w[n:].reshape(n // 16, 16, k)
    wp = torch.cat([wv, wg], dim=1).reshape(n2, k).contiguous()
    bp = None
    if b is not None:
        bp = torch.cat([b[:n].reshape(n // 16, 16), b[n:].reshape(n // 16, 16)], dim=1).reshape(n2).contiguous()
    return wp, bp


# ------------------------------------------------------------------------------------------------
# ops
# ------------------------------------------------------------------------------------------------

def groupnorm(x1, gamma, beta, groups, eps, silu, x2=None, frames_per_stat=1):
    """x1 (N, HW, C1) [, x2 (N, HW, C2)] -> (N, HW, C1+C2): GroupNorm(+SiLU) of the channel concat.
    frames_per_stat = f: statistics over f consecutive images (nn.GroupNorm on the 5-D tensor, inference_v1.yaml)."""
    lib = L.load()
    _req(x1, F16, "x1")
    N, HW, C1 = x1.shape
    C2 = 0
    if x2 is not None:
        _req(x2, F16, "x2")
        C2 = x2.shape[-1]
    Ctot = C1 + C2
    y = torch.empty((N, HW, Ctot), dtype=F16, device=x1.device)
    ws = torch.empty((lib.anip_groupnorm_ws_floats(N, HW, Ctot, groups),), dtype=F32, device=x1.device)
    if _WORK is not None:
        if frames_per_stat == 1 and lib.anip_groupnorm_single_launch(N, HW, Ctot, groups):
            _work(K_GN_APPLY, N * HW * Ctot * 4, f"N{N} HW{HW} C{Ctot} silu{int(bool(silu))} slab")
        else:
            _work(K_GN_STATS, N * HW * Ctot * 2, f"N{N} HW{HW} C{Ctot}")
            _work(K_GN_APPLY, N * HW * Ctot * 4, f"N{N} HW{HW} C{Ctot} silu{int(bool(silu))}")
    L.check(lib.anip_groupnorm_frames(_p(x1), C1, _p(x2), C2, _p(_req(gamma, F32, "gamma")), _p(_req(beta, F32, "beta")),
                                      _p(y), N, HW, groups, float(eps), int(bool(silu)), int(frames_per_stat), _p(ws),
                                      _stream()), "anip_groupnorm_frames")
    return y


def rowgemm320_supported(M, C, rows_per_frame=0):
    return bool(L.load().anip_rowgemm320_supported(int(M), int(C), int(rows_per_frame)))


def groupnorm_scale_shift(x, gamma, beta, groups, eps):
    """per-frame GroupNorm statistics of x (N, HW, C) finalised into the affine form (N, C, 2) fp32 = (rstd gamma,
    beta - mean rstd gamma) for affine_linear320"""
    lib = L.load()
    _req(x, F16, "x")
    N, HW, Cc = x.shape
    ws = torch.empty((lib.anip_groupnorm_ws_floats(N, HW, Cc, groups),), dtype=F32, device=x.device)
    out = torch.empty((N, Cc, 2), dtype=F32, device=x.device)
    _work(K_GN_STATS, N * HW * Cc * 2, f"N{N} HW{HW} C{Cc}")
    _work(K_GN_APPLY, N * Cc * 8, f"N{N} C{Cc} scale_shift")
    L.check(lib.anip_groupnorm_scale_shift(_p(x), _p(_req(gamma, F32, "gamma")), _p(_req(beta, F32, "beta")), _p(out), N, HW,
                                           Cc, groups, float(eps), _p(ws), _stream()), "anip_groupnorm_scale_shift")
    return out


def affine_linear320(x, scale_shift, rows_per_frame, W, bias=None):
    """(x * scale[frame] + shift[frame]) W^T + bias at C = 320 -> 320 (anip_affine_linear320): x (M, 320) fp16, scale_shift
    (frames, 320, 2) fp32, W (320, 320) fp16"""
    lib = L.load()
    _req(x, F16, "x")
    _req(W, F16, "W")
    M, Cc = x.shape
    assert tuple(W.shape) == (Cc, Cc) and scale_shift.shape[0] * rows_per_frame == M
    out = torch.empty_like(x)
    _work(K_GEMM, 2 * M * Cc * Cc, f"affine_linear M{M} C{Cc}", M * Cc * 2 * 2 + Cc * Cc * 2)
    L.check(lib.anip_affine_linear320(_p(x), _p(_req(scale_shift, F32, "scale_shift")), int(rows_per_frame), _p(W), _p(bias),
                                      _p(out), M, Cc, _stream()), "anip_affine_linear320")
    return out


def ln_qkv_projection(x, gamma, beta, w_qkv, heads, q_alpha, eps=1e-5):
    """LayerNorm(x) -> (q token-major x q_alpha (M, C), k head-major (heads, M, d), v^T (C, M)) in one launch
    (anip_ln_qkv_projection; C = 320, 8 heads): w_qkv (3C, C) fp16 = [to_q; to_k; to_v]"""
    lib = L.load()
    _req(x, F16, "x")
    _req(w_qkv, F16, "w_qkv")
    M, Cc = x.shape
    d = Cc // heads
    assert tuple(w_qkv.shape) == (3 * Cc, Cc)
    q = torch.empty((M, Cc), dtype=F16, device=x.device)
    k = torch.empty((heads, M, d), dtype=F16, device=x.device)
    vt = torch.empty((Cc, M), dtype=F16, device=x.device)
    _work(K_GEMM, 2 * M * 3 * Cc * Cc, f"ln_qkv M{M} C{Cc}", M * Cc * 2 * 4 + 3 * Cc * Cc * 2)
    L.check(lib.anip_ln_qkv_projection(_p(x), _p(_req(gamma, F32, "gamma")), _p(_req(beta, F32, "beta")), float(eps),
                                       _p(w_qkv), _p(q), float(q_alpha), _p(k), _p(vt), M, M, Cc, heads, _stream()),
            "anip_ln_qkv_projection")
    return q, k, vt


def layernorm(x, gamma, beta, eps=1e-5, pe=None, rows_per_frame=0, frames=0):
    lib = L.load()
    _req(x, F16, "x")
    M, Cc = x.shape
    y = torch.empty_like(x)
    _work(K_LAYERNORM, M * Cc * 4, f"M{M} C{Cc}")
    L.check(lib.anip_layernorm(_p(x), _p(_req(gamma, F32, "gamma")), _p(_req(beta, F32, "beta")), _p(y), M, Cc,
                               float(eps), _p(pe), int(rows_per_frame), int(frames), _stream()), "anip_layernorm")
    return y


def gemm(A, W, bias=None, A2=None, rowbias=None, rows_per_group=0, residual=None, act=0, out_f32=False,
         alpha=1.0, out=None, conv=None, batch=1, ldo=None, ldr=None, trans_out=False, head_dim=0, debug_ws=None):
    """out = epilogue(alpha * A @ W^T).

    A (M, K) fp16 [+ A2 (M, K2): K split over two sources]; W (N, K) fp16; bias (N,) fp32;
    rowbias (M/rows_per_group, N) fp32 (row stride = rowbias.stride(0): a column slice of a wider
    table is fine); residual (M, N) fp16; act=1 -> GEGLU (W packed by pack_geglu); act=2 -> quick-GELU
    x * sigmoid(1.702 x) on (alpha acc + bias + rowbias), ahead of the residual (CLIP's MLP).
    A / W may be column slices of wider row-major matrices (row stride = .stride(-2)).
    conv: dict(Nimg, Hin, Win, Cin, Hout, Wout, stride, pad, upsample) -> A is the NHWC image batch.
    batch > 1: A (B, M, K) or (M, K) shared; W (B, N, K) or (N, K) shared -> out (B, M, N).
    trans_out: return the transposed result (N, M) (bias only, fp16).
    head_dim > 0: head-major result (N / head_dim, M, head_dim): each head's rows contiguous (K of ref_attention).
    """
    lib = L.load()
    p = L.GemmParams()
    for t, nm in ((A, "A"), (W, "W")):
        if not t.is_cuda:
            raise L.HipLibraryError(f"{nm}: expected a GPU tensor (the hot path has no CPU fallback)")
        if t.dtype != F16:
            raise TypeError(f"{nm}: expected fp16, got {t.dtype}")
        if t.stride(-1) != 1:
            raise ValueError(f"{nm}: innermost dimension must be contiguous")
    batched = A.dim() == 3 or W.dim() == 3
    if batched:
        M, K = A.shape[-2:]
        N = W.shape[-2]
        p.batch = batch
        p.strideA = A.stride(0) if A.dim() == 3 else 0
        p.strideW = W.stride(0) if W.dim() == 3 else 0
        p.strideO = M * N
        p.lda = A.stride(-2)
    elif conv is not None:
        _req(A, F16, "A")
        M = conv["Nimg"] * conv["Hout"] * conv["Wout"]
        K = 9 * conv["Cin"]
        N = W.shape[0]
        p.conv = int(conv.get("korder", 1))
        for k in ("Nimg", "Hin", "Win", "Cin", "Hout", "Wout", "stride", "pad"):
            setattr(p, k, int(conv[k]))
        p.upsample = int(bool(conv.get("upsample", False)))
        p.batch = 1
    else:
        M, K1 = A.shape
        K = K1
        N = W.shape[0]
        p.lda = A.stride(0)
        p.batch = 1
        if A2 is not None:
            if A2.dtype != F16 or A2.stride(-1) != 1:
                raise TypeError("A2: expected fp16 with contiguous rows")
            p.A2, p.lda2, p.K1 = _p(A2), A2.stride(0), K1
            K = K1 + A2.shape[1]
    assert W.shape[-1] == K, f"W has K={W.shape[-1]}, expected {K}"
    n_out = N // 2 if act == 1 else N
    if trans_out:
        assert not batched and act == 0 and not out_f32 and residual is None and rowbias is None
        if out is None:
            out = torch.empty((N, M), dtype=F16, device=A.device)
        p.trans_out = 1
        ldo = M if ldo is None else ldo
    if head_dim:
        assert not batched and conv is None and act == 0 and not out_f32 and not trans_out and N % head_dim == 0
        if out is None:
            out = torch.empty((N // head_dim, M, head_dim), dtype=F16, device=A.device)
        assert out.numel() == M * N and out.is_contiguous()
        p.head_dim = int(head_dim)
    if out is None:
        shape = (batch, M, n_out) if batched else (M, n_out)
        out = torch.empty(shape, dtype=F32 if out_f32 else F16, device=A.device)
    p.A, p.W, p.ldw = _p(A), _p(W), W.stride(-2)
    p.out, p.ldo, p.out_f32 = _p(out), int(ldo if ldo is not None else n_out), int(out_f32)
    p.M, p.N, p.K = M, N, K
    p.alpha = float(alpha)
    p.bias = _p(bias)
    if rowbias is not None:
        p.rowbias, p.rows_per_group, p.ld_rowbias = _p(rowbias), int(rows_per_group), rowbias.stride(0)
    if residual is not None:
        if residual.dtype != F16:
            raise TypeError("residual: expected fp16")
        p.residual, p.ldr = _p(residual), int(ldr if ldr is not None else n_out)
    p.act = int(act)
    if _WORK is not None:
        if conv is not None:
            desc = (f"N{conv['Nimg']} {conv['Hin']}x{conv['Win']} Cin{conv['Cin']} Cout{N} s{conv['stride']}"
                    f"{' up' if conv.get('upsample') else ''}{' rb' if rowbias is not None else ''}"
                    f"{' res' if residual is not None else ''}")
        else:
            desc = (f"M{M} N{N} K{K}{' b%d' % batch if batched else ''}{' geglu' if act == 1 else ''}"
                    f"{' A2' if A2 is not None else ''}{' rb' if rowbias is not None else ''}"
                    f"{' res' if residual is not None else ''}{' T' if trans_out else ''}{' f32' if out_f32 else ''}"
                    f"{' hm' if head_dim else ''}")
        nb = max(1, int(p.batch))
        a_bytes = (conv["Nimg"] * conv["Hin"] * conv["Win"] * conv["Cin"] if conv is not None else
                   M * K * (nb if (not batched or A.dim() == 3) else 1)) * 2
        w_bytes = N * K * 2 * (nb if (batched and W.dim() == 3) else 1)
        o_bytes = M * n_out * nb * (4 if out_f32 else 2) + (M * n_out * 2 if residual is not None else 0)
        _work(K_CONV3X3 if conv is not None else K_GEMM, 2 * M * N * K * nb, desc, a_bytes + w_bytes + o_bytes)
    if M <= 16384:   # few output tiles: the library may want fp32 scratch for split-K
        wsb = lib.anip_gemm_workspace_bytes(C.byref(p))
        if wsb > 0:
            ws = torch.empty((wsb,), dtype=torch.uint8, device=A.device)
            p.workspace, p.workspace_bytes = _p(ws), wsb
    if debug_ws is not None:     # experiment builds only (segment-timing kernels write their counters here)
        p.workspace, p.workspace_bytes = _p(debug_ws), debug_ws.numel() * debug_ws.element_size()
    L.check(lib.anip_gemm(C.byref(p), _stream()), "anip_gemm")
    return out


def ffn_geglu(x, w1p, b1p, w2, b2, residual=None):
    """Fused GEGLU feed-forward (anip_ffn_geglu; C = 320 only; the engine's default there): x (M, C) fp16, w1p / b1p packed by
    pack_geglu, w2 (C, 4C) fp16, b2 (C,) fp32, residual (M, C) fp16 -> (M, C) fp16."""
    lib = L.load()
    _req(x, F16, "x")
    _req(w1p, F16, "w1p")
    _req(w2, F16, "w2")
    M, Cc = x.shape
    assert tuple(w1p.shape) == (8 * Cc, Cc) and tuple(w2.shape) == (Cc, 4 * Cc)
    out = torch.empty_like(x)
    _work(K_GEMM, 2 * M * Cc * (8 * Cc) + 2 * M * Cc * (4 * Cc), f"ffn_geglu M{M} C{Cc}",
          M * Cc * 2 * (3 if residual is not None else 2) + 12 * Cc * Cc * 2)
    L.check(lib.anip_ffn_geglu(_p(x), _p(w1p), _p(_req(b1p, F32, "b1p")), _p(w2), _p(b2), _p(residual), _p(out), M, Cc,
                               _stream()), "anip_ffn_geglu")
    return out


def ffn_geglu_ln(x, gamma, beta, w1p, b1p, w2, b2, residual=None, eps=1e-5):
    """residual + FeedForward(LayerNorm(x)) in one launch (anip_ffn_geglu_ln; C = 320): x (M, C) fp16 RAW rows, gamma / beta (C,)
    fp32, the rest as ffn_geglu"""
    lib = L.load()
    _req(x, F16, "x")
    _req(w1p, F16, "w1p")
    _req(w2, F16, "w2")
    M, Cc = x.shape
    assert tuple(w1p.shape) == (8 * Cc, Cc) and tuple(w2.shape) == (Cc, 4 * Cc)
    out = torch.empty_like(x)
    _work(K_GEMM, 2 * M * Cc * (8 * Cc) + 2 * M * Cc * (4 * Cc), f"ffn_geglu_ln M{M} C{Cc}",
          M * Cc * 2 * 2 + 12 * Cc * Cc * 2)
    L.check(lib.anip_ffn_geglu_ln(_p(x), _p(_req(gamma, F32, "gamma")), _p(_req(beta, F32, "beta")), float(eps), _p(w1p),
                                  _p(_req(b1p, F32, "b1p")), _p(w2), _p(b2), _p(residual), _p(out), M, Cc, _stream()),
            "anip_ffn_geglu_ln")
    return out


def conv3x3(x, Wp, bias, stride=1, pad=1, upsample=False, pad_hi=None, rowbias=None, rows_per_group=0,
            residual=None, out_f32=False, korder=None, debug_ws=None):
    """x (N, H, W, Cin) fp16, Wp (Cout, 9*Cin) packed by pack_conv3x3 -> (N, Ho, Wo, Cout).
    pad = low-side padding; pad_hi (default = pad) = high-side padding (VAE encoder uses 0/1).
    korder: the K order Wp was packed in (default: pack_conv3x3's default for this Cin)."""
    N, H, Wd, Cin = x.shape
    if pad_hi is None:
        pad_hi = pad
    He, We = (2 * H, 2 * Wd) if upsample else (H, Wd)
    Ho = (He + pad + pad_hi - 3) // stride + 1
    Wo = (We + pad + pad_hi - 3) // stride + 1
    conv = dict(Nimg=N, Hin=H, Win=Wd, Cin=Cin, Hout=Ho, Wout=Wo, stride=stride, pad=pad, upsample=upsample,
                korder=conv_korder(Cin) if korder is None else korder)
    res2 = residual.reshape(-1, Wp.shape[0]) if residual is not None else None
    out = gemm(x, Wp, bias, rowbias=rowbias, rows_per_group=rows_per_group, residual=res2, conv=conv,
               out_f32=out_f32, debug_ws=debug_ws)
    return out.reshape(N, Ho, Wo, Wp.shape[0])


def pack_conv_direct(w):
    """torch conv weight [Cout, Cin, k, k] -> [k*k*Cin, Cout8] (tap-major, output channel fastest, Cout padded
    with zeros to a multiple of 8): the LDS image of anip_conv_direct."""
    co, ci, kh, kw = w.shape
    co8 = (co + 7) // 8 * 8
    wp = torch.zeros((kh * kw * ci, co8), dtype=w.dtype, device=w.device)
    wp[:, :co] = w.permute(2, 3, 1, 0).reshape(kh * kw * ci, co)
    return wp.contiguous()


def conv_direct(x, wp, bias, Cout, ksize, stride=1, pad=1, relu=False, residual=None):
    """x (N, H, W, Cin) fp16, wp from pack_conv_direct [, residual (N, Ho, Wo, Cout)] -> (N, Ho, Wo, Cout) fp16."""
    lib = L.load()
    _req(x, F16, "x")
    _req(wp, F16, "wp")
    N, H, Wd, Cin = x.shape
    assert wp.shape[0] == ksize * ksize * Cin and wp.shape[1] == (Cout + 7) // 8 * 8
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (Wd + 2 * pad - ksize) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), dtype=F16, device=x.device)
    _work(K_CONV_SMALL, x.numel() * 2 + y.numel() * 2, f"direct N{N} {H}x{Wd} Cin{Cin} Cout{Cout} k{ksize} s{stride}")
    if residual is not None:
        _req(residual, F16, "residual")
        assert residual.numel() == y.numel()
    L.check(lib.anip_conv_direct(_p(x), _p(wp), _p(bias), _p(residual), _p(y), N, H, Wd, Cin, Cout, ksize, stride, pad,
                                 int(bool(relu)), _stream()), "anip_conv_direct")
    return y


def batchnorm(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, relu=True):
    """x (M, C) fp16 channels-last rows -> relu?(BatchNorm(x)); batch statistics unless running stats are given."""
    lib = L.load()
    _req(x, F16, "x")
    M, Cc = x.shape
    y = torch.empty_like(x)
    ws = torch.empty((lib.anip_batchnorm_ws_floats(M, Cc),), dtype=F32, device=x.device)
    _work(K_BATCHNORM, M * Cc * (6 if running_mean is None else 4), f"M{M} C{Cc}")
    L.check(lib.anip_batchnorm(_p(x), _p(_req(gamma, F32, "gamma")), _p(_req(beta, F32, "beta")), _p(running_mean),
                               _p(running_var), _p(y), M, Cc, float(eps), int(bool(relu)), _p(ws), _stream()),
            "anip_batchnorm")
    return y


def ref_attention(q, ldq, k, ldk, vt, ldvt, n_frames, T, heads, d, kref=None, ldkr=0, vtref=None, ldvtr=0,
                  ref_index=None, scale=None, n_ref_frames=0, k_head_stride=0, kref_head_stride=0, q_log2_scaled=False,
                  frame_mod=0):
    """see anip_ref_attention_ex; returns (n_frames*T, heads*d) fp16.  frame_mod = m > 0: q / k / vt hold m frames, frame n
    attends with those of frame n % m (ANIP_ATTN_FRAME_MOD).  `n_ref_frames` (frames whose
    ref_index >= 0) is only used for the profiler's FLOP count.  k / kref head-major (gemm(head_dim=d) output, shape
    (heads, tokens, d)): ldk = d and k_head_stride = tokens * d.  q_log2_scaled: q already carries scale * log2(e)
    (ANIP_ATTN_Q_LOG2_SCALED: gemm(..., alpha=attn_q_alpha(d)) of the to_q projection)."""
    lib = L.load()
    frame_mod = int(frame_mod)
    if not 0 <= frame_mod < 32768:      # packed into bits 16.. of a C int: a larger value would wrap silently
        raise ValueError(f"ref_attention: frame_mod must be in [0, 32767], got {frame_mod}")
    held = frame_mod if frame_mod else n_frames
    if frame_mod and n_frames % frame_mod:
        raise ValueError(f"ref_attention: n_frames ({n_frames}) must be a multiple of frame_mod ({frame_mod})")
    if q.numel() < held * T * heads * d or k.numel() < held * T * heads * d or vt.numel() < held * T * heads * d:
        raise ValueError(f"ref_attention: q / k / vt must hold {held} frames of {T} tokens x {heads * d} channels "
                         f"(q {tuple(q.shape)}, k {tuple(k.shape)}, vt {tuple(vt.shape)})")
    _work(K_REF_ATTN, 4 * T * T * heads * d * (n_frames + (n_ref_frames if ref_index is not None else 0)),
          f"Nf{n_frames} T{T} h{heads} d{d} ref{n_ref_frames if ref_index is not None else 0}",
          2 * heads * d * (4 * n_frames * T + (2 * kref.numel() // (heads * d) if kref is not None else 0)))
    out = torch.empty((n_frames * T, heads * d), dtype=F16, device=q.device)
    if scale is None:
        scale = d ** -0.5
    L.check(lib.anip_ref_attention_ex(_p(q), ldq, _p(k), ldk, _p(vt), ldvt, _p(kref), ldkr, _p(vtref), ldvtr,
                                      _p(ref_index), _p(out), heads * d, n_frames, T, heads, d, float(scale),
                                      int(k_head_stride), int(kref_head_stride),
                                      (1 if q_log2_scaled else 0) | (int(frame_mod) << 16), _stream()),
            "anip_ref_attention")
    return out


def attn_q_alpha(d, scale=None):
    """alpha of the to_q projection GEMM for ref_attention(q_log2_scaled=True): softmax scale x log2(e)"""
    return float((d ** -0.5 if scale is None else scale) * 1.4426950408889634)


def temporal_attention(qkv, B, F, T, heads, d, scale=None):
    lib = L.load()
    _req(qkv, F16, "qkv")
    out = torch.empty((B * F * T, heads * d), dtype=F16, device=qkv.device)
    if scale is None:
        scale = d ** -0.5
    _work(K_TEMPORAL_ATTN, B * F * T * heads * d * 4 * 2, f"B{B} F{F} T{T} h{heads} d{d}")
    L.check(lib.anip_temporal_attention(_p(qkv), _p(out), B, F, T, heads, d, float(scale), _stream()),
            "anip_temporal_attention")
    return out


def pack_temporal_qkv(wq, wk, wv):
    """[to_q; to_k; to_v] of a C = 320 motion-module attention (each (320, 320) fp16) in the row order of
    anip_temporal_qkv_attention: per head PAIR p its 80 to_q rows, 80 to_k rows, 80 to_v rows"""
    assert tuple(wq.shape) == tuple(wk.shape) == tuple(wv.shape) == (320, 320)
    parts = []
    for p_ in range(4):
        for w in (wq, wk, wv):
            parts.append(w[80 * p_:80 * p_ + 80])
    return torch.cat(parts, dim=0).to(F16).contiguous()


def temporal_qkv_attention_supported(F, T, C, heads):
    return bool(L.load().anip_temporal_qkv_attention_supported(int(F), int(T), int(C), int(heads)))


def temporal_qkv_attention(x, gamma, beta_pe, w_packed, B, F, T, heads, eps=1e-5, scale=None):
    """LayerNorm(+pe) -> to_q / to_k / to_v -> temporal self-attention in one launch (anip_temporal_qkv_attention; F = 16,
    C = 320, 8 heads): x (B*F*T, C) fp16, gamma (C,) fp32, beta_pe (F, C) fp32 = norm bias + pe[frame], w_packed from
    pack_temporal_qkv -> attention output (B*F*T, C) fp16 (the input of to_out)."""
    lib = L.load()
    _req(x, F16, "x")
    _req(w_packed, F16, "w_packed")
    M, Cc = x.shape
    d = Cc // heads
    assert M == B * F * T and tuple(w_packed.shape) == (3 * Cc, Cc) and tuple(beta_pe.shape) == (F, Cc)
    if scale is None:
        scale = d ** -0.5
    out = torch.empty_like(x)
    _work(K_GEMM, 2 * M * 3 * Cc * Cc + 4 * M * F * Cc, f"tqkv_attn M{M} C{Cc} F{F}", M * Cc * 2 * 2 + 3 * Cc * Cc * 2)
    L.check(lib.anip_temporal_qkv_attention(_p(x), _p(_req(gamma, F32, "gamma")), _p(_req(beta_pe, F32, "beta_pe")),
                                            _p(w_packed), _p(out), B, F, T, Cc, heads, float(eps), float(scale), _stream()),
            "anip_temporal_qkv_attention")
    return out


def softmax_rows(s):
    lib = L.load()
    _req(s, F32, "s")
    rows = s.numel() // s.shape[-1]
    p = torch.empty(s.shape, dtype=F16, device=s.device)
    _work(K_SOFTMAX, s.numel() * 6, f"rows{rows} cols{s.shape[-1]}")
    L.check(lib.anip_softmax_rows(_p(s), _p(p), rows, s.shape[-1], _stream()), "anip_softmax_rows")
    return p


def linear_small(x, W, bias=None, silu_in=False, out=None):
    """x (M<=16, K) fp32, W (N, K) fp16 -> (M, N) fp32 (written into `out` when given)."""
    lib = L.load()
    _req(x, F32, "x")
    _req(W, F16, "W")
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty((M, N), dtype=F32, device=x.device) if out is None else _req(out, F32, "out")
    assert tuple(y.shape) == (M, N)
    _work(K_LINEAR_SMALL, N * K * 2, f"M{M} N{N} K{K}")
    L.check(lib.anip_linear_small(_p(x), _p(W), _p(bias), _p(y), M, N, K, int(bool(silu_in)), _stream()),
            "anip_linear_small")
    return y


def add(a, b):
    lib = L.load()
    _req(a, F16, "a")
    _req(b, F16, "b")
    out = torch.empty_like(a)
    _work(K_ELEMENTWISE, a.numel() * 6, "add")
    L.check(lib.anip_add(_p(a), _p(b), _p(out), a.numel(), _stream()), "anip_add")
    return out


def window_accumulate(pred, acc, counter, frames, S, Fw, L_, HWC):
    lib = L.load()
    _work(K_ELEMENTWISE, S * Fw * HWC * 10, "window_accumulate")
    L.check(lib.anip_window_accumulate(_p(pred), _p(acc), _p(counter), _p(frames), S, Fw, L_, HWC, _stream()),
            "anip_window_accumulate")


def cfg_ddim_step(acc, counter, latents, latents_f16, S, L_, HWC, guidance, sa, sb, sap, sbp):
    lib = L.load()
    _work(K_ELEMENTWISE, L_ * HWC * (4 * S + 10), "cfg_ddim_step")
    L.check(lib.anip_cfg_ddim_step(_p(acc), _p(counter), _p(latents), _p(latents_f16), S, L_, HWC, float(guidance),
                                   float(sa), float(sb), float(sap), float(sbp), _stream()), "anip_cfg_ddim_step")


def ncfhw_to_nhwc(src):
    """(B, C, F, H, W) fp32/fp16 -> (B*F, H, W, C) fp16."""
    lib = L.load()
    if not src.is_cuda:
        raise L.HipLibraryError("ncfhw_to_nhwc: expected a GPU tensor")
    src = src.contiguous()
    B, Cc, Fr, H, Wd = src.shape
    dst = torch.empty((B * Fr, H, Wd, Cc), dtype=F16, device=src.device)
    _work(K_ELEMENTWISE, src.numel() * (src.element_size() + 2), "ncfhw_to_nhwc")
    L.check(lib.anip_ncfhw_to_nhwc(_p(src), int(src.dtype == F32), _p(dst), B, Cc, Fr, H * Wd, _stream()),
            "anip_ncfhw_to_nhwc")
    return dst


def nhwc_to_ncfhw(src, B, out_f32=False, scale=1.0, shift=0.0, clamp01=False):
    """(B*F, H, W, C) fp16 -> (B, C, F, H, W)."""
    lib = L.load()
    _req(src, F16, "src")
    BF, H, Wd, Cc = src.shape
    Fr = BF // B
    dst = torch.empty((B, Cc, Fr, H, Wd), dtype=F32 if out_f32 else F16, device=src.device)
    _work(K_ELEMENTWISE, src.numel() * (2 + dst.element_size()), "nhwc_to_ncfhw")
    L.check(lib.anip_nhwc_to_ncfhw(_p(src), _p(dst), int(out_f32), B, Cc, Fr, H * Wd, float(scale), float(shift),
                                   int(bool(clamp01)), _stream()), "anip_nhwc_to_ncfhw")
    return dst


def u8_to_f16(src, scale=1.0, shift=0.0):
    """uint8 tensor -> fp16 tensor of the same shape: scale * x + shift."""
    lib = L.load()
    if not src.is_cuda:
        raise L.HipLibraryError("u8_to_f16: expected a GPU tensor")
    if src.dtype != torch.uint8 or not src.is_contiguous():
        raise TypeError("u8_to_f16: expected a contiguous uint8 tensor")
    dst = torch.empty(src.shape, dtype=F16, device=src.device)
    _work(K_ELEMENTWISE, src.numel() * 3, "u8_to_f16")
    L.check(lib.anip_u8_to_f16(_p(src), _p(dst), src.numel(), float(scale), float(shift), _stream()), "anip_u8_to_f16")
    return dst


def f16_to_u8(src, scale=1.0, shift=0.0):
    """fp16 tensor -> uint8 tensor of the same shape: trunc(255 * fp16(clamp(scale * x + shift, 0, 1)))."""
    lib = L.load()
    _req(src, F16, "src")
    dst = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
    _work(K_ELEMENTWISE, src.numel() * 3, "f16_to_u8")
    L.check(lib.anip_f16_to_u8(_p(src), _p(dst), src.numel(), float(scale), float(shift), _stream()), "anip_f16_to_u8")
    return dst
