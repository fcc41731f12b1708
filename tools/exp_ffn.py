"""fused feed-forward vs the two-GEMM path at the 64x64 level (M = 131072, C = 320)"""
import json, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import hipops as ops
DEV = "cuda"
def r16(*s, scale=1.0): return (torch.randn(s, device=DEV) * scale).half()
def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M, C = 131072, 320
x, R = r16(M, C), r16(M, C)
W1, b1 = r16(8 * C, C, scale=C ** -0.5), torch.randn(8 * C, device=DEV)
W2, b2 = r16(C, 4 * C, scale=(4 * C) ** -0.5), torch.randn(C, device=DEV)
w1p, b1p = ops.pack_geglu(W1, b1)
two = timeit(lambda: ops.gemm(ops.gemm(x, w1p, b1p, act=1), W2, b2, residual=R))
fused = timeit(lambda: ops.ffn_geglu(x, w1p, b1p, W2, b2, R))
d = (ops.ffn_geglu(x, w1p, b1p, W2, b2, R).float() - ops.gemm(ops.gemm(x, w1p, b1p, act=1), W2, b2, residual=R).float()).abs().max().item()
print(json.dumps(dict(two_gemm_us=round(two, 1), fused_us=round(fused, 1), max_abs_diff=d)))
