"""Runs the hot kernels of the C2 workload (512x512, L=16, CFG: 32 frames) a few times each, at the shapes and with the
epilogues the pipeline uses, for rocprofv3 --pmc passes (tools/pmc_round.sh; summarised by tools/pmc_summarize.py).
The first kernel is an fp16 add over 1 GiB — a known byte count (2 reads + 1 write per element, far beyond the
256 MiB Infinity Cache) that calibrates FETCH_SIZE / WRITE_SIZE in this access width.
usage: python tools/pmc_kernels.py [calib|attn|gemm|conv|norm|all]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import hipops as ops  # noqa: E402

DEV = "cuda"
NF = 32
REP = 3


def r16(*shape, scale=1.0):
    return (torch.randn(shape, device=DEV) * scale).half()


def calib():
    n = 512 * 1024 * 1024          # elements: 1 GiB per operand
    a = torch.ones(n, dtype=torch.float16, device=DEV)
    b = torch.ones(n, dtype=torch.float16, device=DEV)
    for _ in range(REP):
        ops.add(a, b)
    torch.cuda.synchronize()


def attn(T=4096, heads=8, d=40):
    """the engine's operand layouts: Q token-major and pre-multiplied by d^-1/2 log2 e, K head-major, V^T"""
    C = heads * d
    q = (r16(NF * T, C).float() * ops.attn_q_alpha(d)).half()
    k = r16(heads, NF * T, d)
    vt = r16(C, NF * T)
    kref, vtref = r16(heads, 2 * T, d), r16(C, 2 * T)
    ridx = torch.tensor([-1] * (NF // 2) + [1] * (NF - NF // 2), dtype=torch.int32, device=DEV)
    for _ in range(REP):
        ops.ref_attention(q, C, k, d, vt, NF * T, NF, T, heads, d, kref=kref, ldkr=d, vtref=vtref, ldvtr=2 * T,
                          ref_index=ridx, k_head_stride=NF * T * d, kref_head_stride=2 * T * d, q_log2_scaled=True)
    torch.cuda.synchronize()


def gemm(M, N, K, geglu=False, res=False):
    A, W = r16(M, K), r16(N, K, scale=K ** -0.5)
    b = torch.randn(N, device=DEV)
    if geglu:
        W, b = ops.pack_geglu(W, b)
    R = r16(M, N) if res else None
    for _ in range(REP):
        ops.gemm(A, W, b, act=1 if geglu else 0, residual=R)
    torch.cuda.synchronize()


def conv(N, H, Cin, Cout, res=True):
    x = r16(N, H, H, Cin)
    w = ops.pack_conv3x3(r16(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5))
    b = torch.randn(Cout, device=DEV)
    R = r16(N, H, H, Cout) if res else None
    for _ in range(REP):
        ops.conv3x3(x, w, b, residual=R)
    torch.cuda.synchronize()


def norm():
    x = r16(NF, 4096, 320)
    g, b = torch.ones(320, device=DEV), torch.zeros(320, device=DEV)
    for _ in range(REP):
        ops.groupnorm(x, g, b, 32, 1e-5, True)
    x2 = r16(NF * 4096, 320)
    for _ in range(REP):
        ops.layernorm(x2, g, b)
    qkv = r16(NF * 4096, 960)
    for _ in range(REP):
        ops.temporal_attention(qkv, 2, 16, 4096, 8, 40)
    torch.cuda.synchronize()


def fused():
    """round 5's fused kernels at the 64x64 level: LayerNorm(+pe) -> q|k|v -> temporal attention, LayerNorm -> FFN"""
    T, C = 4096, 320
    x = r16(NF * T, C, scale=1.5)
    g, b = 1 + 0.1 * torch.randn(C, device=DEV), 0.1 * torch.randn(C, device=DEV)
    bpe = (b[None] + 0.5 * torch.randn(16, C, device=DEV)).contiguous()
    wp = ops.pack_temporal_qkv(*(r16(C, C, scale=C ** -0.5) for _ in range(3)))
    for _ in range(REP):
        ops.temporal_qkv_attention(x, g, bpe, wp, 2, 16, T, 8)
    w1p, b1p = ops.pack_geglu(r16(8 * C, C, scale=C ** -0.5), torch.randn(8 * C, device=DEV))
    W2, b2 = r16(C, 4 * C, scale=(4 * C) ** -0.5), torch.randn(C, device=DEV)
    for _ in range(REP):
        ops.ffn_geglu_ln(x, g, b, w1p, b1p, W2, b2, x)
    torch.cuda.synchronize()


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("fused", "all"):
    fused()
if what in ("calib", "all"):
    calib()
if what in ("attn", "all"):
    attn()
    attn(1024, 8, 80)
if what in ("gemm", "all"):
    gemm(NF * 4096, 320, 320, res=True)          # out-proj / proj_out 64^2
    gemm(NF * 4096, 960, 320)                    # temporal qkv 64^2
    gemm(NF * 4096, 2560, 320, geglu=True)       # ff-in 64^2
    gemm(NF * 4096, 320, 1280, res=True)         # ff-out 64^2
    gemm(NF * 1024, 640, 640, res=True)          # out-proj 32^2
    gemm(NF * 1024, 5120, 640, geglu=True)       # ff-in 32^2
    gemm(8192, 8192, 8192)
if what in ("conv", "all"):
    conv(NF, 64, 320, 320)
    conv(NF, 32, 640, 640)
if what in ("norm", "all"):
    norm()
