"""HBM write bandwidth of the epilogue's store pattern vs contiguous bytes per row of one wave instruction
(tools/exp_store_pattern.hip, compiled here with hipcc).  Usage on the GPU box: python tools/exp_store_pattern.py"""
import ctypes
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = "/tmp/exp_store_pattern.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO,
                       os.path.join(HERE, "exp_store_pattern.hip")])
lib = ctypes.CDLL(SO)
lib.exp_store.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def run(M, N, R, with_res, iters=20):
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    res = torch.randn(M, N, device="cuda").half() if with_res else None
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.exp_store(out.data_ptr(), res.data_ptr() if with_res else None, M, N, R, st)  # noqa: E731
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    by = M * N * 2 * (2 if with_res else 1)
    print(json.dumps(dict(M=M, N=N, rows_per_instr=R, bytes_per_row=1024 // R, residual=with_res, us=best * 1e3,
                          tbps=by / best / 1e9)), flush=True)


for (M, N) in ((131072, 1024), (131072, 256), (32768, 2048)):
    for with_res in (False, True):
        for R in (16, 8, 4, 2):
            run(M, N, R, with_res)
