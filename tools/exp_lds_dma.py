"""Global -> LDS (LDS-DMA) throughput of one CU vs access shape and queue depth (tools/exp_lds_dma.hip, compiled here).
Usage on the GPU box: python tools/exp_lds_dma.py      -> one JSON line per case: bytes per cycle per CU, TB/s over the chip"""
import ctypes
import json
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = "/tmp/exp_lds_dma.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO,
                       os.path.join(HERE, "exp_lds_dma.hip")])
lib = ctypes.CDLL(SO)
lib.exp_dma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                        ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
props = torch.cuda.get_device_properties(0)
CUS = props.multi_processor_count
ROUNDS = 400


def run(per_wave, row_bytes, ld_bytes, span, swizzle, waves_active=8):
    src = torch.empty(CUS * span, dtype=torch.uint8, device="cuda").random_(0, 255)
    cyc = torch.zeros(CUS * 8, dtype=torch.int64, device="cuda")
    sink = torch.zeros(CUS, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(2):
        if k == 1:
            e0.record()
        rc = lib.exp_dma(src.data_ptr(), cyc.data_ptr(), sink.data_ptr(), CUS, ROUNDS, per_wave, row_bytes, ld_bytes, span, swizzle,
                         waves_active, st)
        assert rc == 0, rc
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = cyc.double().median().item() / ROUNDS
    by = waves_active * per_wave * 1024
    print(json.dumps(dict(per_wave=per_wave, waves=waves_active, row_bytes=row_bytes, ld_bytes=ld_bytes, span_kb=span >> 10,
                          swizzle=swizzle, cycles_per_round=round(c, 1), cycles_per_instr=round(c / per_wave, 1),
                          bytes_per_cycle_per_cu=round(by / c, 2), chip_tbps=round(by * ROUNDS * CUS / (ms * 1e-3) / 1e12, 2))),
          flush=True)


for span in (256 << 10, 8 << 20):                       # L2-resident per CU / streaming (2 GiB over the chip)
    for row_bytes, ld in ((128, 640), (128, 2560), (64, 640), (1024, 1024)):
        for per_wave in (1, 2, 4, 9):
            run(per_wave, row_bytes, ld, span, 1 if row_bytes <= 128 else 0)
run(4, 128, 640, 256 << 10, 0)                           # no swizzle
run(9, 128, 640, 256 << 10, 1, waves_active=4)           # four issuing waves (one wave group of the role-alternating loop)
