"""Micro-benchmark of round 5's fused kernels against the launches they replace, at the 64x64 level of BASELINE C2
(M = 32 frames x 4096 pixels = 131072 rows, C = 320).  One JSON line per case (us warm / us cache-cold).
Usage (GPU box): python tools/exp_fused_blocks.py"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import tools.bench_kernels as bk  # noqa: E402
from aniportrait_amd import hipops as ops  # noqa: E402

DEV = "cuda"
B, Fr, T, C, heads, d = 2, 16, 4096, 320, 8, 40
M = B * Fr * T


def both(fn):
    bk.COLD = False
    w = bk.timeit(fn) * 1e6
    bk.COLD = True
    c = bk.timeit(fn) * 1e6
    bk.COLD = False
    return round(w, 1), round(c, 1)


def main():
    torch.manual_seed(0)
    x = (torch.randn(M, C, device=DEV) * 1.5).half()
    gamma = 1 + 0.1 * torch.randn(C, device=DEV)
    beta = 0.1 * torch.randn(C, device=DEV)
    pe = 0.5 * torch.randn(Fr, C, device=DEV)
    wq, wk, wv, wo = ((torch.randn(C, C, device=DEV) * C ** -0.5).half() for _ in range(4))
    bo = torch.randn(C, device=DEV)
    wcat = torch.cat([wq, wk, wv]).contiguous()
    wp = ops.pack_temporal_qkv(wq, wk, wv)
    bpe = (beta[None] + pe).contiguous()
    flops = 2 * M * 3 * C * C

    def three():
        nh = ops.layernorm(x, gamma, beta, pe=pe, rows_per_frame=T, frames=Fr)
        return ops.temporal_attention(ops.gemm(nh, wcat), B, Fr, T, heads, d)

    def fused():
        return ops.temporal_qkv_attention(x, gamma, bpe, wp, B, Fr, T, heads)

    for name, fn in (("temporal_front_three_launches", three), ("temporal_front_fused", fused)):
        w, c = both(fn)
        print(json.dumps({"case": name, "us": w, "us_cold": c, "tflops": round(flops / w * 1e-6, 1)}), flush=True)
    a = fused()
    w, c = both(lambda: ops.gemm(a, wo, bo, residual=x))
    print(json.dumps({"case": "temporal_to_out_residual", "us": w, "us_cold": c}), flush=True)

    W1 = (torch.randn(8 * C, C, device=DEV) * C ** -0.5).half()
    b1 = torch.randn(8 * C, device=DEV)
    W2 = (torch.randn(C, 4 * C, device=DEV) * (4 * C) ** -0.5).half()
    b2 = torch.randn(C, device=DEV)
    w1p, b1p = ops.pack_geglu(W1, b1)
    fflops = 2 * M * C * 8 * C + 2 * M * C * 4 * C
    for name, fn in (("layernorm+ffn_geglu", lambda: ops.ffn_geglu(ops.layernorm(x, gamma, beta), w1p, b1p, W2, b2, x)),
                     ("ffn_geglu_ln", lambda: ops.ffn_geglu_ln(x, gamma, beta, w1p, b1p, W2, b2, x))):
        w, c = both(fn)
        print(json.dumps({"case": name, "us": w, "us_cold": c, "tflops": round(fflops / w * 1e-6, 1)}), flush=True)
    # norm1 -> q | k | v^T of the spatial block
    qa = ops.attn_q_alpha(d)

    def three_proj():
        nh = ops.layernorm(x, gamma, beta)
        return ops.gemm(nh, wq, alpha=qa), ops.gemm(nh, wk, head_dim=d), ops.gemm(nh, wv, trans_out=True)

    for name, fn in (("layernorm+q+k+vt", three_proj), ("ln_qkv_projection", lambda: ops.ln_qkv_projection(x, gamma, beta, wcat, heads, qa))):
        w, c = both(fn)
        print(json.dumps({"case": name, "us": w, "us_cold": c, "tflops": round(flops / w * 1e-6, 1)}), flush=True)
    # GroupNorm -> proj_in
    x3 = x.reshape(B * Fr, T, C)
    for name, fn in (("groupnorm+proj_in", lambda: ops.gemm(ops.groupnorm(x3, gamma, beta, 32, 1e-6, False).reshape(M, C), wo, bo)),
                     ("gn_scale_shift+affine_linear320", lambda: ops.affine_linear320(x, ops.groupnorm_scale_shift(x3, gamma, beta, 32, 1e-6), T, wo, bo))):
        w, c = both(fn)
        print(json.dumps({"case": name, "us": w, "us_cold": c, "tflops": round(2 * M * C * C / w * 1e-6, 1)}), flush=True)


if __name__ == "__main__":
    main()
