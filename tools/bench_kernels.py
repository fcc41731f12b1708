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


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def r16(*shape, scale=1.0):
    return (torch.randn(shape, device=DEV) * scale).half()


def bench_gemm(M, N, K, tag):
    A, W = r16(M, K), r16(N, K, scale=K ** -0.5)
    b = torch.randn(N, device=DEV)
    t = timeit(lambda: ops.gemm(A, W, b))
    print(json.dumps(dict(kernel="gemm", tag=tag, M=M, N=N, K=K, ms=t * 1e3, tflops=2 * M * N * K / t / 1e12)), flush=True)


def bench_conv(N, H, Cin, Cout, tag, up=False, stride=1):
    x = r16(N, H, H, Cin)
    w = ops.pack_conv3x3(r16(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5))
    b = torch.randn(Cout, device=DEV)
    t = timeit(lambda: ops.conv3x3(x, w, b, stride=stride, upsample=up))
    Ho = (2 * H if up else H) // stride
    fl = 2 * N * Ho * Ho * Cout * 9 * Cin
    print(json.dumps(dict(kernel="conv3x3", tag=tag, N=N, H=H, Cin=Cin, Cout=Cout, ms=t * 1e3, tflops=fl / t / 1e12)), flush=True)


def bench_attn(Nf, T, heads, d, tag):
    C = heads * d
    qk = r16(Nf * T, 2 * C)
    vt = r16(C, Nf * T)
    kref, vtref = r16(2 * T, C), r16(C, 2 * T)
    ridx = torch.tensor([-1] * (Nf // 2) + [1] * (Nf - Nf // 2), dtype=torch.int32, device=DEV)
    t = timeit(lambda: ops.ref_attention(qk, 2 * C, qk[:, C:], 2 * C, vt, Nf * T, Nf, T, heads, d, kref=kref, ldkr=C,
                                         vtref=vtref, ldvtr=2 * T, ref_index=ridx))
    fl = 4 * T * T * C * (Nf // 2) + 4 * T * 2 * T * C * (Nf - Nf // 2)
    print(json.dumps(dict(kernel="ref_attention", tag=tag, Nf=Nf, T=T, d=d, ms=t * 1e3, tflops=fl / t / 1e12)), flush=True)


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
    quick = "--quick" in sys.argv
    print(json.dumps(dict(device=ops.device_info())), flush=True)
    NF = 32
    if "--attn" in sys.argv:
        bench_attn(NF, 4096, 8, 40, "64^2 d40")
        bench_attn(NF, 1024, 8, 80, "32^2 d80")
        bench_attn(NF, 256, 8, 160, "16^2 d160")
        bench_attn(NF, 64, 8, 160, "8^2 d160")
        return
    # linear layers of the 64^2 / 32^2 / 16^2 levels
    bench_gemm(NF * 4096, 640, 320, "qk-proj 64^2")
    bench_gemm(NF * 4096, 320, 320, "out-proj 64^2")
    bench_gemm(NF * 4096, 2560, 320, "ff-in 64^2 (as plain)")
    bench_gemm(NF * 4096, 320, 1280, "ff-out 64^2")
    bench_gemm(NF * 1024, 640, 2560, "ff-out 32^2")
    bench_gemm(NF * 256, 1280, 5120, "ff-out 16^2")
    bench_gemm(8192, 8192, 8192, "square 8k")
    if not quick:
        bench_gemm(4096, 4096, 4096, "square 4k")
        bench_gemm(NF * 64, 1280, 1280, "proj 8^2")
    bench_conv(NF, 64, 320, 320, "res 64^2 320")
    bench_conv(NF, 32, 640, 640, "res 32^2 640")
    bench_conv(NF, 16, 1280, 1280, "res 16^2 1280")
    bench_conv(NF, 8, 2560, 1280, "res 8^2 2560->1280")
    if not quick:
        bench_conv(NF, 64, 960, 320, "res 64^2 960->320")
        bench_conv(16, 256, 256, 256, "vae 256^2 256")
        bench_conv(16, 512, 128, 128, "vae 512^2 128")
        bench_conv(NF, 32, 640, 640, "up 32->64", up=True)
    bench_attn(NF, 4096, 8, 40, "64^2 d40")
    bench_attn(NF, 1024, 8, 80, "32^2 d80")
    bench_attn(NF, 256, 8, 160, "16^2 d160")
    bench_gn(NF, 4096, 320, "64^2 C320")
    bench_gn(NF, 4096, 960, "64^2 C960")
    bench_gn(16, 262144, 128, "vae 512^2 C128")
    bench_ln(NF * 4096, 320, "64^2 C320")
    bench_ln(NF * 256, 1280, "16^2 C1280")
    bench_temporal(2, 16, 4096, 8, 40, "64^2 d40")
    bench_temporal(2, 16, 256, 8, 160, "16^2 d160")


if __name__ == "__main__":
    main()
