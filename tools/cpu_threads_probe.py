"""How many torch threads should bench.py's cpu_baseline use under the GPU box's cgroup quota?  Times the oracle's 256x256,
L=4 sample (one UNet3D call timed by itself) at each thread count given on the command line.  `python tools/cpu_threads_probe.py 16 64`"""
import importlib.util
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
for n in [int(a) for a in sys.argv[1:]] or [16, 64]:
    r = bench.cpu_baseline(size=256, frames=4, c1=(64, 2, 1), steps_timed=1, threads=n)
    print(json.dumps({"threads": n, "quota": r["cgroup_cpu_quota"], "fixed_s": r["fixed_seconds"], "step_s": r["step_seconds"],
                      "vae_frame_s": r["vae_frame_seconds"], "cpu_tflops": r["cpu_tflops"]}), flush=True)
