"""experiment: is the reference-attention kernel bound by the strided (token-major, head-interleaved) K reads?
Same FLOPs / blocks / tiles two ways: (a) the pipeline's layout — 8 heads interleaved in 640-B token rows, every
80-B K row of a head at a 1280-B stride; (b) heads folded into the frame dimension with contiguous 80-B K rows."""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import hipops as ops  # noqa: E402
from tools.bench_kernels import timeit, r16  # noqa: E402

DEV = "cuda"
T, d = 4096, 40


def run(heads, Nf, tag):
    C = heads * d
    q = r16(Nf * T, C)
    k = r16(Nf * T, C)
    qk = torch.cat([q, k], dim=1).contiguous()
    vt = r16(C, Nf * T)
    nref = Nf // 16                       # same ratio as 2 reference samples for 32 frames
    kref, vtref = r16(nref * T, C), r16(C, nref * T)
    ridx = torch.tensor([-1] * (Nf // 2) + [nref - 1] * (Nf - Nf // 2), dtype=torch.int32, device=DEV)
    for name, (qq, ldq, kk, ldk) in {"fused q|k rows": (qk, 2 * C, qk[:, C:], 2 * C), "separate q, k": (q, C, k, C)}.items():
        t = timeit(lambda: ops.ref_attention(qq, ldq, kk, ldk, vt, Nf * T, Nf, T, heads, d, kref=kref, ldkr=C, vtref=vtref,
                                             ldvtr=nref * T, ref_index=ridx))
        fl = 4 * T * T * C * (Nf // 2) + 4 * T * 2 * T * C * (Nf - Nf // 2)
        print(json.dumps(dict(tag=tag, layout=name, heads=heads, Nf=Nf, k_row_stride_bytes=ldk * 2, ms=t * 1e3,
                              tflops=fl / t / 1e12)), flush=True)


run(8, 32, "pipeline layout")
run(1, 256, "heads folded into frames (contiguous K rows)")


def run_pad(pad_v, pad_k, tag):
    """power-of-two leading dimensions: V^T rows are 2*Nf*T bytes apart (256 KiB), the reference's 16 KiB"""
    heads, Nf = 8, 32
    C = heads * d
    q = r16(Nf * T, C)
    kbuf = r16(Nf * T, C + pad_k)
    k = kbuf[:, :C]
    vbuf = r16(C, Nf * T + pad_v)
    vt = vbuf[:, : Nf * T]
    kref = r16(2 * T, C)
    vrbuf = r16(C, 2 * T + pad_v)
    vtref = vrbuf[:, : 2 * T]
    ridx = torch.tensor([-1] * (Nf // 2) + [1] * (Nf - Nf // 2), dtype=torch.int32, device=DEV)
    t = timeit(lambda: ops.ref_attention(q, C, k, C + pad_k, vt, Nf * T + pad_v, Nf, T, heads, d, kref=kref, ldkr=C, vtref=vtref,
                                         ldvtr=2 * T + pad_v, ref_index=ridx))
    fl = 4 * T * T * C * (Nf // 2) + 4 * T * 2 * T * C * (Nf - Nf // 2)
    print(json.dumps(dict(tag=tag, pad_v=pad_v, pad_k=pad_k, ms=t * 1e3, tflops=fl / t / 1e12)), flush=True)


run_pad(0, 0, "separate q,k; ld V^T = 2^17")
run_pad(64, 0, "V^T rows padded by 128 B")
run_pad(192, 0, "V^T rows padded by 384 B")
run_pad(64, 64, "V^T and K rows padded by 128 B")
