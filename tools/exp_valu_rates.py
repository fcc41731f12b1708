"""Issue rates of the VALU / transcendental / MFMA instructions behind the kernel models of DESIGN.md
(tools/exp_valu_rates.hip, compiled here with hipcc).  Usage on the GPU box: python tools/exp_valu_rates.py
Prints, per instruction and waves per SIMD, the cycles per instruction per wave and per SIMD."""
import ctypes
import json
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = "/tmp/exp_valu_rates.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO,
                       os.path.join(HERE, "exp_valu_rates.hip")])
lib = ctypes.CDLL(SO)
lib.exp_rate.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4

OPS = {0: ("v_exp_f32", 8), 1: ("v_rcp_f32", 8), 2: ("v_fma_f32", 8), 3: ("v_pk_fma_f16", 8), 4: ("v_dot2_f32_f16", 8),
       5: ("v_cvt_pkrtz_f16_f32", 8), 6: ("v_max3_f32", 8), 7: ("v_mov_b32_dpp row_ror:8", 8), 8: ("v_exp_f16", 8),
       9: ("v_pk_mul_f32", 8), 10: ("v_mfma_f32_16x16x32_f16", 8), 11: ("v_mfma_f32_32x32x16_f16", 8),
       12: ("softmax mix: 8 fma + 8 exp + 4 cvt_pk", 20), 13: ("v_permlane32_swap_b32", 8), 14: ("v_permlane16_swap_b32", 8),
       15: ("v_pk_fma_f32", 8),
       # same-wave MFMA / VALU overlap: cycles per ITERATION of the mix (per_iter = 1)
       16: ("ITER: 2 mfma32x32x16 + 8 v_exp", 1), 17: ("ITER: 2 mfma32x32x16 + 16 v_fma", 1),
       18: ("ITER: 4 mfma16x16x32 + 8 v_exp", 1),
       19: ("ITER: attention unit mix (6 mfma32 + 12 mfma16 | 48 fma 32 exp 16 cvt 16 max3)", 1),
       20: ("ITER: that VALU multiset alone", 1), 21: ("ITER: that MFMA multiset alone", 1),
       22: ("v_mfma_f32_32x32x8_f16 (legacy K=8)", 8), 23: ("v_mfma_f32_16x16x16_f16 (legacy K=16)", 8),
       24: ("v_max3_f32 DEPENDENT chain", 8),
       25: ("ITER: planned unit, 14 mfma32 (VGPR acc) | 32 exp 16 cvt 16 max3 8 fma", 1),
       26: ("ITER: planned unit, PV accumulators in AGPRs", 1), 27: ("ITER: planned unit, all accumulators in AGPRs", 1),
       28: ("ITER: planned unit's VALU multiset alone", 1), 29: ("ITER: 2 mfma32x32x16 (AGPR acc) + 16 v_fma", 1),
       30: ("ITER: PHASED unit, independent: 6 mfma | 72 VALU | 8 mfma(AGPR)", 1),
       31: ("ITER: PHASED unit with the data dependences score -> softmax -> P V", 1)}
ONLY = [int(x) for x in os.environ.get("RATES_ONLY", "").split(",") if x]
ITERS = 2000
CUS = torch.cuda.get_device_properties(0).multi_processor_count
for op, (name, per_iter) in OPS.items():
    if ONLY and op not in ONLY:
        continue
    row = {"instruction": name}
    for w in (1, 2, 4):
        threads = 256 * w
        out = torch.empty(CUS * threads, dtype=torch.float32, device="cuda")
        cyc = torch.zeros(CUS * threads // 64, dtype=torch.int64, device="cuda")
        rt = torch.zeros(CUS * threads // 64, dtype=torch.int64, device="cuda")
        for _ in range(2):
            rc = lib.exp_rate(op, CUS, threads, ITERS, out.data_ptr(), cyc.data_ptr(), rt.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.exp_rate(op, CUS, threads, ITERS, out.data_ptr(), cyc.data_ptr(), rt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        wall_us = e0.elapsed_time(e1) * 1e3
        row.setdefault("ticks_per_us", {})[w] = round(cyc.double().median().item() / wall_us, 1)   # s_memtime ticks per microsecond of wall time
        # s_memrealtime: constant 100 MHz -> real nanoseconds per iteration per SIMD and the s_memtime tick rate seen by a wave
        ns = rt.double().median().item() * 10.0
        row.setdefault("ns_per_iter_per_simd", {})[w] = round(ns / ITERS / w, 2)
        row.setdefault("ticks_per_us_in_wave", {})[w] = round(cyc.double().median().item() / (ns * 1e-3), 1)
        c = cyc.double().median().item() / (ITERS * per_iter)
        row[f"{w}_waves_per_simd"] = {"cycles_per_instr_per_wave": round(c, 2), "cycles_per_instr_per_simd": round(c / w, 2)}
    print(json.dumps(row), flush=True)
