"""Runs the dominant kernels of the C2 workload a few times each (for rocprofv3 --pmc passes).
usage: python tools/pmc_kernels.py [attn|gemm|conv|all]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import hipops as ops  # noqa: E402

DEV = "cuda"
NF = 32


def r16(*shape, scale=1.0):
    return (torch.randn(shape, device=DEV) * scale).half()


def attn(T=4096, heads=8, d=40):
    C = heads * d
    qk = r16(NF * T, 2 * C)
    vt = r16(C, NF * T)
    kref, vtref = r16(2 * T, C), r16(C, 2 * T)
    ridx = torch.tensor([-1] * (NF // 2) + [1] * (NF - NF // 2), dtype=torch.int32, device=DEV)
    for _ in range(3):
        ops.ref_attention(qk, 2 * C, qk[:, C:], 2 * C, vt, NF * T, NF, T, heads, d, kref=kref, ldkr=C, vtref=vtref,
                          ldvtr=2 * T, ref_index=ridx)
    torch.cuda.synchronize()


def gemm(M, N, K, geglu=False):
    A, W = r16(M, K), r16(N, K, scale=K ** -0.5)
    b = torch.randn(N, device=DEV)
    if geglu:
        W, b = ops.pack_geglu(W, b)
    for _ in range(3):
        ops.gemm(A, W, b, act=1 if geglu else 0)
    torch.cuda.synchronize()


def conv(N, H, Cin, Cout):
    x = r16(N, H, H, Cin)
    w = ops.pack_conv3x3(r16(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5))
    b = torch.randn(Cout, device=DEV)
    for _ in range(3):
        ops.conv3x3(x, w, b)
    torch.cuda.synchronize()


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("attn", "all"):
    attn()
if what in ("gemm", "all"):
    gemm(NF * 4096, 640, 320)
    gemm(NF * 4096, 2560, 320, geglu=True)
    gemm(NF * 4096, 320, 1280)
    gemm(NF * 1024, 5120, 640, geglu=True)
    gemm(8192, 8192, 8192)
if what in ("conv", "all"):
    conv(NF, 64, 320, 320)
    conv(NF, 32, 640, 640)
