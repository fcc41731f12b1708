"""GEMM decomposition experiment: time a few C2 shapes under ANIP_GEMM2_DBG (bit 1: no epilogue, 2: no
global->LDS DMA, 4: no MFMA).  usage: ANIP_GEMM2_DBG=k python tools/exp_gemm.py"""
import json, os, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import hipops as ops

DEV = "cuda"
def r16(*s, scale=1.0): return (torch.randn(s, device=DEV) * scale).half()
def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
dbg = os.environ.get("ANIP_GEMM2_DBG", "0")
res = {}
for (M, N, K, geglu, resid) in [(131072, 2560, 320, True, False), (131072, 2560, 320, False, False), (131072, 320, 1280, False, True),
                         (131072, 320, 320, False, True), (131072, 640, 320, False, False), (32768, 5120, 640, True, False),
                         (8192, 8192, 8192, False, False)]:
    A, W = r16(M, K), r16(N, K, scale=K ** -0.5)
    b = torch.randn(N, device=DEV)
    if geglu: W, b = ops.pack_geglu(W, b)
    R = r16(M, N) if resid else None
    out = torch.empty((M, N // 2 if geglu else N), dtype=torch.float16, device=DEV)
    us = timeit(lambda: ops.gemm(A, W, b, act=1 if geglu else 0, residual=R, out=out))
    res[f"{M}x{N}x{K}{' geglu' if geglu else ''}{' res' if resid else ''}"] = round(us, 1)
x = r16(32, 64, 64, 320); w = ops.pack_conv3x3(r16(320, 320, 3, 3, scale=(9 * 320) ** -0.5)); b = torch.randn(320, device=DEV)
res["conv 64^2 320"] = round(timeit(lambda: ops.conv3x3(x, w, b)), 1)
print(json.dumps({"dbg": dbg, "us": res}))
res2 = {}
for (M, N, K) in [(2048, 1280, 5120), (2048, 1280, 1280), (2048, 3840, 1280)]:
    A, W = r16(M, K), r16(N, K, scale=K ** -0.5); b = torch.randn(N, device=DEV); R = r16(M, N)
    res2[f"{M}x{N}x{K} res"] = round(timeit(lambda: ops.gemm(A, W, b, residual=R)), 1)
for (Cin, H) in [(1280, 8), (2560, 8), (1280, 16)]:
    x = r16(32, H, H, Cin); w = ops.pack_conv3x3(r16(1280, Cin, 3, 3, scale=(9 * Cin) ** -0.5)); b = torch.randn(1280, device=DEV)
    res2[f"conv {H}^2 {Cin}->1280"] = round(timeit(lambda: ops.conv3x3(x, w, b)), 1)
print(json.dumps({"dbg": dbg, "small_M_us": res2}))
