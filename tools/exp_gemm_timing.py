"""Segment timing of the quarter-phased wide-tile main loop (experiment build: python -m aniportrait_amd.build
-DANIP_GEMM2_TIMING --out=libaniportrait_hip_timing.so; run with ANIP_LIB=<that .so> ANIP_GEMM2_SCHED=2).
Per sub-step j and wave group: cycles of LOAD (up to lgkmcnt(0)), DMA wait (vmcnt), barrier wait, COMPUTE (16-20 MFMAs),
barrier wait — averaged per K-tile, from waves 0 / 4 of the middle block (s_memtime: the instrumentation itself costs
about 10 %)."""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import hipops as ops  # noqa: E402

DEV = "cuda"


def r16(*shape, scale=1.0):
    return (torch.randn(shape, device=DEV) * scale).half()


def report(tag, ws, us):
    w = ws.cpu().view(2, 32)
    out = {"tag": tag, "us": us}
    for g in range(2):
        nk = int(w[g, 20])
        a = [float(x) / max(nk, 1) for x in w[g, :20]]
        rows = []
        for j in range(4):
            rows.append(dict(j=j, load=a[16 + j] if j in (0, 3) else a[4 * j], dma_wait=a[4 * j] if j in (0, 3) else 0.0,
                             bar1=a[4 * j + 1], compute=a[4 * j + 2], bar2=a[4 * j + 3]))
        out[f"group{g}"] = {"k_tiles": nk, "cycles_per_k_tile": sum(a[:16]) + a[16] + a[19], "substeps": rows}
    print(json.dumps(out), flush=True)
    for g in range(2):
        r = out[f"group{g}"]
        print(f"  {tag} group {g}: {r['cycles_per_k_tile']:.0f} cycles / K-tile ({r['k_tiles']} tiles); " + " | ".join(
            f"j{x['j']}: L {x['load']:.0f} dma {x['dma_wait']:.0f} b {x['bar1']:.0f} C {x['compute']:.0f} b {x['bar2']:.0f}" for x in r["substeps"]),
            flush=True)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 5 * 1e3


def gemm_case(M, N, K, tag, res=False):
    A, W = r16(M, K), r16(N, K, scale=K ** -0.5)
    b = torch.randn(N, device=DEV)
    R = r16(M, N) if res else None
    ws = torch.zeros(64, dtype=torch.int32, device=DEV)
    us = timed(lambda: ops.gemm(A, W, b, residual=R, debug_ws=ws))
    report(tag, ws, us)


def conv_case(N, H, Cin, Cout, tag):
    x = r16(N, H, H, Cin)
    w = ops.pack_conv3x3(r16(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5))
    b = torch.randn(Cout, device=DEV)
    R = r16(N, H, H, Cout)
    ws = torch.zeros(64, dtype=torch.int32, device=DEV)
    us = timed(lambda: ops.conv3x3(x, w, b, residual=R, debug_ws=ws))
    report(tag, ws, us)


if __name__ == "__main__":
    gemm_case(8192, 8192, 8192, "gemm 8k^3")
    gemm_case(32 * 1024, 640, 2560, "gemm 32^2 ff-out M32768 N640 K2560", res=True)
    gemm_case(32 * 4096, 960, 320, "gemm 64^2 temporal qkv M131072 N960 K320")
    conv_case(32, 64, 320, 320, "conv 64^2 320->320")
    conv_case(32, 32, 640, 640, "conv 32^2 640->640")
