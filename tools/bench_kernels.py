"""Micro-benchmarks of the hot kernels at the BASELINE C2 shapes (512x512, L=16, CFG => 32 frames).
Prints one JSON line per case: achieved TFLOP/s (dense-contraction FLOPs) or GB/s (algorithmic bytes).
Usage (GPU box): python tools/bench_kernels.py [--quick]"""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import hipops as ops  # noqa: E402

DEV = "cuda"


COLD = False          # --cold: evict the Infinity Cache / L2 before every timed launch
BOTH = False          # --both: GEMM rows carry the warm time (us) and the cache-cold time (us_cold)
_FLUSH = None


def _flush():
    """write 768 MB: more than the 256 MB Infinity Cache + 8 x 4 MB L2, so the next launch finds its operands in HBM —
    the pessimistic end of what a kernel sees inside the pipeline (there, operands written one or two launches earlier are
    partly still cached; the default warm loop is the optimistic end: e.g. the N = K = 320 residual layers run 57 us warm,
    79 us in the pipeline)"""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)
    _FLUSH.add_(1)


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if COLD:
        tot = 0.0
        for _ in range(iters):
            _flush()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / iters * 1e-3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def r16(*shape, scale=1.0):
    import os
    if os.environ.get("OPERANDS_ZERO"):   # all-zero operands: same instruction stream, no data toggling (clock / power check)
        return torch.zeros(shape, device=DEV, dtype=torch.float16)
    return (torch.randn(shape, device=DEV) * scale).half()


def bench_gemm(M, N, K, tag, res=False, rb=False, geglu=False, a2=0, trans=False):
    """the epilogue variants of the path: bias / +residual / +row-group bias (time embedding, collapsed attn2) /
    GEGLU / two-source A (fused skip concat) / transposed out (V^T projection)"""
    A, W = r16(M, K - a2), r16(N, K, scale=K ** -0.5)
    A2 = r16(M, a2) if a2 else None
    b = None if trans else torch.randn(N, device=DEV)
    if geglu:
        W, b = ops.pack_geglu(W, b)
    n_out = N // 2 if geglu else N
    R = r16(M, n_out) if res else None
    RB = torch.randn((M // 4096 if M >= 4096 else 1, n_out), device=DEV) if rb else None
    fn = lambda: ops.gemm(A, W, b, A2=A2, residual=R, rowbias=RB, rows_per_group=4096 if M >= 4096 else M,  # noqa: E731
                          act=1 if geglu else 0, trans_out=trans)
    t = timeit(fn)
    t_cold = None
    if BOTH:
        global COLD
        COLD = True
        t_cold = timeit(fn) * 1e6
        COLD = False
    by = 2 * (A.numel() + (A2.numel() if a2 else 0) + W.numel() + M * n_out * (2 if res else 1))
    print(json.dumps(dict(kernel="gemm", tag=tag, M=M, N=N, K=K, res=res, rb=rb, geglu=geglu, a2=a2, trans=trans,
                          us=t * 1e6, tflops=2 * M * N * K / t / 1e12, gbps=by / t / 1e9, us_cold=t_cold)), flush=True)


def bench_conv(N, H, Cin, Cout, tag, up=False, stride=1, res=False, rb=False, pad=1, pad_hi=None):
    x = r16(N, H, H, Cin)
    w = ops.pack_conv3x3(r16(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5))
    b = torch.randn(Cout, device=DEV)
    y = ops.conv3x3(x, w, b, stride=stride, upsample=up, pad=pad, pad_hi=pad_hi)
    R = torch.randn_like(y) if res else None
    RB = torch.randn((2, Cout), device=DEV) if rb else None
    rpg = (y.shape[0] * y.shape[1] * y.shape[2]) // 2
    t = timeit(lambda: ops.conv3x3(x, w, b, stride=stride, upsample=up, pad=pad, pad_hi=pad_hi, residual=R, rowbias=RB,
                                   rows_per_group=rpg))
    fl = 2 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * 9 * Cin
    print(json.dumps(dict(kernel="conv3x3", tag=tag, N=N, H=H, Cin=Cin, Cout=Cout, res=res, rb=rb, up=up, stride=stride,
                          us=t * 1e6, tflops=fl / t / 1e12)), flush=True)


def bench_attn(Nf, T, heads, d, tag):
    """operand layouts of engine.transformer_block: Q token-major, K head-major (heads, tokens, d), V^T (C, tokens)"""
    C = heads * d
    q = (r16(Nf * T, C).float() * ops.attn_q_alpha(d)).half()     # what gemm(alpha=attn_q_alpha(d)) of to_q delivers
    k = r16(heads, Nf * T, d)
    vt = r16(C, Nf * T)
    kref, vtref = r16(heads, 2 * T, d), r16(C, 2 * T)
    ridx = torch.tensor([-1] * (Nf // 2) + [1] * (Nf - Nf // 2), dtype=torch.int32, device=DEV)
    import os
    if os.environ.get("ATTN_ZERO"):       # all-zero operands: same instruction stream, no data toggling (clock / power check)
        for x in (q, k, vt, kref, vtref):
            x.zero_()
        tag += " zeros"
    t = timeit(lambda: ops.ref_attention(q, C, k, d, vt, Nf * T, Nf, T, heads, d, kref=kref, ldkr=d, vtref=vtref,
                                         ldvtr=2 * T, ref_index=ridx, k_head_stride=Nf * T * d,
                                         kref_head_stride=2 * T * d, q_log2_scaled=True))
    fl = 4 * T * T * C * (Nf // 2) + 4 * T * 2 * T * C * (Nf - Nf // 2)
    print(json.dumps(dict(kernel="ref_attention", tag=tag, Nf=Nf, T=T, d=d, ms=t * 1e3, us=t * 1e6, tflops=fl / t / 1e12)), flush=True)


def bench_gn(N, HW, C, tag):
    x = r16(N, HW, C)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    t = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, True))
    by = 3 * x.numel() * 2  # read twice, write once
    print(json.dumps(dict(kernel="groupnorm_silu", tag=tag, N=N, HW=HW, C=C, ms=t * 1e3, gbps=by / t / 1e9)), flush=True)


def bench_ln(M, C, tag):
    x = r16(M, C)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    t = timeit(lambda: ops.layernorm(x, g, b))
    print(json.dumps(dict(kernel="layernorm", tag=tag, M=M, C=C, ms=t * 1e3, gbps=2 * x.numel() * 2 / t / 1e9)), flush=True)


def bench_temporal(B, F, T, heads, d, tag):
    qkv = r16(B * F * T, 3 * heads * d)
    t = timeit(lambda: ops.temporal_attention(qkv, B, F, T, heads, d))
    by = qkv.numel() * 2 * 4 / 3
    print(json.dumps(dict(kernel="temporal_attention", tag=tag, T=T, d=d, ms=t * 1e3, gbps=by / t / 1e9)), flush=True)


def main():
    """--only gemm,conv,attn,norm   restrict the families;   --cold   cache-cold launches (see _flush);   env switches of the library (read once per process) make
    one invocation = one kernel variant: ANIP_LIB=<experiment build>; OPERANDS_ZERO=1 / ATTN_ZERO=1: all-zero operands (clock / power check)."""
    import os
    global COLD, BOTH
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only"):
            only = set(a.split("=", 1)[1].split(","))
        if a == "--cold":
            COLD = True
        if a == "--both":      # GEMM rows: warm loop in `us`, cache-cold launches in `us_cold`
            BOTH = True
    want = lambda k: only is None or k in only  # noqa: E731
    print(json.dumps(dict(device=ops.device_info(), cold=COLD,
                          env={k: v for k, v in os.environ.items() if k.startswith("ANIP_")})), flush=True)
    NF = 32
    if want("gemm"):
        # the Linear / 1x1 layers of one UNet3D call at C2 (shape, epilogue) — launches per call in the tag
        bench_gemm(NF * 4096, 320, 320, "64^2 out-proj/proj_out x4", res=True)
        bench_gemm(NF * 4096, 320, 320, "64^2 attn1 out-proj +attn2 x1", res=True, rb=True)
        bench_gemm(NF * 4096, 320, 320, "64^2 proj_in x2")
        bench_gemm(NF * 4096, 320, 320, "64^2 to_v^T x1", trans=True)
        bench_gemm(NF * 4096, 640, 320, "64^2 q|k x1")
        bench_gemm(NF * 4096, 960, 320, "64^2 temporal qkv x2")
        bench_gemm(NF * 4096, 2560, 320, "64^2 ff-in geglu x2", geglu=True)
        bench_gemm(NF * 4096, 320, 1280, "64^2 ff-out x2", res=True)
        bench_gemm(NF * 4096, 320, 640, "64^2 shortcut 640->320 (concat)", a2=320)
        bench_gemm(NF * 1024, 640, 640, "32^2 out-proj x4", res=True)
        bench_gemm(NF * 1024, 1920, 640, "32^2 temporal qkv x2")
        bench_gemm(NF * 1024, 5120, 640, "32^2 ff-in geglu x2", geglu=True)
        bench_gemm(NF * 1024, 640, 2560, "32^2 ff-out x2", res=True)
        bench_gemm(NF * 256, 1280, 1280, "16^2 out-proj x4", res=True)
        bench_gemm(NF * 256, 3840, 1280, "16^2 temporal qkv x2")
        bench_gemm(NF * 256, 10240, 1280, "16^2 ff-in geglu x2", geglu=True)
        bench_gemm(NF * 256, 1280, 5120, "16^2 ff-out x2", res=True)
        bench_gemm(NF * 256, 1280, 1280, "16^2 proj_in / to_q", )
        bench_gemm(NF * 256, 1280, 1280, "16^2 to_v^T", trans=True)
        bench_gemm(NF * 64, 1280, 1280, "8^2 out-proj", res=True)
        bench_gemm(NF * 64, 1280, 1280, "8^2 proj_in / to_q")
        bench_gemm(NF * 64, 3840, 1280, "8^2 temporal qkv")
        bench_gemm(NF * 64, 1280, 2560, "8^2 shortcut 2560->1280 (concat)", a2=1280)
        bench_gemm(NF * 64, 10240, 1280, "8^2 ff-in geglu", geglu=True)
        bench_gemm(NF * 64, 1280, 5120, "8^2 ff-out", res=True)
        bench_gemm(512, 1280, 1280, "refnet 16^2 out-proj", res=True)
        bench_gemm(512, 1280, 5120, "refnet 16^2 ff-out", res=True)
        bench_gemm(128, 1280, 1280, "refnet 8^2 out-proj", res=True)
        bench_gemm(128, 1280, 2560, "refnet 8^2 shortcut (concat)", a2=1280)
        bench_gemm(8192, 8192, 8192, "square 8k")
    if "sk" in (only or ()):   # the tile-starved levels (16x16 / 8x8: the shapes round 6's in-launch split-K experiment was measured on, profiles/r06/h_*)
        bench_gemm(NF * 256, 1280, 1280, "16^2 out-proj x4", res=True)
        bench_gemm(NF * 256, 1280, 1280, "16^2 proj_in / to_q")
        bench_gemm(NF * 256, 3840, 1280, "16^2 temporal qkv x2")
        bench_gemm(NF * 256, 1280, 5120, "16^2 ff-out x2", res=True)
        bench_gemm(NF * 256, 1280, 2560, "16^2 shortcut (concat)", a2=1280)
        bench_gemm(NF * 64, 1280, 1280, "8^2 out-proj", res=True)
        bench_gemm(NF * 64, 3840, 1280, "8^2 temporal qkv")
        bench_gemm(NF * 64, 1280, 5120, "8^2 ff-out", res=True)
        bench_gemm(NF * 64, 1280, 2560, "8^2 shortcut 2560->1280 (concat)", a2=1280)
        bench_gemm(NF * 1024, 640, 640, "32^2 out-proj x4", res=True)
        bench_gemm(NF * 1024, 640, 2560, "32^2 ff-out x2", res=True)
        bench_conv(NF, 16, 1280, 1280, "res 16^2 1280 conv2", res=True)
        bench_conv(NF, 16, 2560, 1280, "res 16^2 2560->1280 conv1", rb=True)
        bench_conv(NF, 8, 1280, 1280, "res 8^2 1280 conv2", res=True)
        bench_conv(NF, 8, 2560, 1280, "res 8^2 2560->1280 conv1", rb=True)
    if want("conv"):
        bench_conv(NF, 64, 320, 320, "res 64^2 320 conv2", res=True)
        bench_conv(NF, 64, 320, 320, "res 64^2 320 conv1", rb=True)
        bench_conv(NF, 64, 640, 320, "res 64^2 640->320 conv1", rb=True)
        bench_conv(NF, 32, 640, 640, "res 32^2 640 conv2", res=True)
        bench_conv(NF, 16, 1280, 1280, "res 16^2 1280 conv2", res=True)
        bench_conv(NF, 8, 1280, 1280, "res 8^2 1280 conv2", res=True)
        bench_conv(NF, 8, 2560, 1280, "res 8^2 2560->1280 conv1", rb=True)
        bench_conv(NF, 32, 640, 640, "up 32->64", up=True)
        bench_conv(2, 8, 1280, 1280, "refnet 8^2 1280 conv2", res=True)
        bench_conv(2, 8, 2560, 1280, "refnet 8^2 2560->1280 conv1", rb=True)
        bench_conv(2, 16, 1280, 1280, "refnet 16^2 1280 conv2", res=True)
        bench_conv(2, 16, 2560, 1280, "refnet 16^2 2560->1280 conv1", rb=True)
        bench_conv(2, 32, 640, 640, "refnet 32^2 640 conv2", res=True)
        bench_conv(16, 256, 256, 256, "vae 256^2 256", res=True)
        bench_conv(16, 512, 128, 128, "vae 512^2 128", res=True)
        bench_conv(1, 256, 256, 256, "vae-enc 256^2 s2 pad(0,1)", stride=2, pad=0, pad_hi=1)
    if want("attn"):
        bench_attn(NF, 4096, 8, 40, "64^2 d40")
        bench_attn(NF, 1024, 8, 80, "32^2 d80")
        bench_attn(NF, 256, 8, 160, "16^2 d160")
        bench_attn(NF, 64, 8, 160, "8^2 d160")
    if want("misc"):   # the small once-per-step / once-per-clip kernels
        x = r16(NF, 64, 64, 4)
        wp = ops.pack_conv_direct(r16(320, 4, 3, 3, scale=1 / 6.0))
        b = torch.randn(320, device=DEV)
        pose = r16(NF, 64, 64, 320)
        t = timeit(lambda: ops.conv_direct(x, wp, b, 320, 3, residual=pose))
        print(json.dumps(dict(kernel="conv_direct", tag="conv_in 4->320 64^2 + pose", us=t * 1e6,
                              gbps=(x.numel() + 2 * pose.numel()) * 2 / t / 1e9)), flush=True)
        xs = torch.randn(2, 1280, device=DEV)
        Wt = r16(20160, 1280, scale=1280 ** -0.5)
        bt = torch.randn(20160, device=DEV)
        t = timeit(lambda: ops.linear_small(xs, Wt, bt, silu_in=True))
        print(json.dumps(dict(kernel="linear_small", tag="stacked time_emb_proj M2 N20160 K1280", us=t * 1e6,
                              gbps=Wt.numel() * 2 / t / 1e9)), flush=True)
        for (n_, hw, c) in ((2, 4096, 320), (1, 4096, 512), (2, 1024, 640), (1, 262144, 128)):
            xg = r16(n_, hw, c)
            g_, b_ = torch.randn(c, device=DEV), torch.randn(c, device=DEV)
            t = timeit(lambda: ops.groupnorm(xg, g_, b_, 32, 1e-5, silu=True))
            print(json.dumps(dict(kernel="groupnorm", tag=f"N{n_} HW{hw} C{c}", us=t * 1e6, gbps=xg.numel() * 6 / t / 1e9)), flush=True)
    if want("norm"):
        bench_gn(NF, 4096, 320, "64^2 C320")
        bench_gn(NF, 4096, 640, "64^2 C640")
        bench_gn(NF, 4096, 960, "64^2 C960")
        bench_gn(NF, 1024, 640, "32^2 C640")
        bench_gn(NF, 1024, 1920, "32^2 C1920")
        bench_gn(NF, 256, 1280, "16^2 C1280")
        bench_gn(NF, 256, 2560, "16^2 C2560")
        bench_gn(NF, 64, 1280, "8^2 C1280")
        bench_gn(NF, 64, 2560, "8^2 C2560")
        bench_gn(16, 262144, 128, "vae 512^2 C128")
        bench_ln(NF * 4096, 320, "64^2 C320")
        bench_ln(NF * 1024, 640, "32^2 C640")
        bench_ln(NF * 256, 1280, "16^2 C1280")
        bench_temporal(2, 16, 4096, 8, 40, "64^2 d40")
        bench_temporal(2, 16, 1024, 8, 80, "32^2 d80")
        bench_temporal(2, 16, 256, 8, 160, "16^2 d160")


if __name__ == "__main__":
    main()
