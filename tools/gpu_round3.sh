#!/bin/bash
# round-3 GPU session driver (one gpurun call = one invocation): `bash tools/gpu_round3.sh <tag> <what...>`
#   parity   fixture-based real-width parity tests (C2 4 steps / C5 768 / L=40 windows) + the rest of test_gpu_real_width
#   ktests   kernel unit tests of the GEMM / conv family under both main-loop schedules
#   bisect   per-block error table hip vs fp32 oracle at 32x32 / 64x64 latents (tests/bisect_parity.py)
#   kbench   tools/bench_kernels.py gemm,conv under ANIP_GEMM2_SCHED=0 and =1
#   bench    the headline bench line (+ per-shape table)
#   pmc      HBM-traffic PMC passes over the eager denoising step, paired with the traced wrapper calls
#   alltests the whole -m gpu suite
TAG=${1:-r03a}; shift
WHAT="$*"
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has parity; then
  echo "== parity (fixtures)"
  timeout 900 python -m pytest tests/test_gpu_real_width.py -m gpu -q -s -k "fixture" > $OUT/parity_fixtures.log 2>&1; echo "rc=$?" >> $OUT/parity_fixtures.log
  grep -E "PSNR|passed|failed|rror|rc=" $OUT/parity_fixtures.log | tail -n 12
fi
if has ktests; then
  for S in 1 0; do
    echo "== kernel tests, ANIP_GEMM2_SCHED=$S"
    ANIP_GEMM2_SCHED=$S timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm or conv or ffn or linear" > $OUT/ktests_sched$S.log 2>&1; echo "rc=$?" >> $OUT/ktests_sched$S.log
    tail -n 4 $OUT/ktests_sched$S.log
  done
fi
if has bisect; then
  echo "== bisect"
  timeout 900 python tests/bisect_parity.py --sizes 32 64 --frames 2 --backends hip --out $OUT/bisect_hip.json > $OUT/bisect_hip.log 2>&1; echo "rc=$?" >> $OUT/bisect_hip.log
  grep -E "conv_out|rc=" $OUT/bisect_hip.log | tail -n 6
fi
if has kbench; then
  for S in 0 1; do
    echo "== kernel bench, ANIP_GEMM2_SCHED=$S"
    ANIP_GEMM2_SCHED=$S timeout 300 python tools/bench_kernels.py --only=gemm,conv > $OUT/kbench_sched$S.jsonl 2>&1; echo "rc=$?"
  done
  python - <<PY
import json
def load(p):
    d={}
    for l in open(p):
        try: r=json.loads(l)
        except Exception: continue
        if "tag" in r: d[(r["kernel"],r["tag"])]=r
    return d
a,b=load("$OUT/kbench_sched0.jsonl"),load("$OUT/kbench_sched1.jsonl")
for k in a:
    if k in b: print("%-8s %-40s sched0 %8.1f us %7.1f TF | sched1 %8.1f us %7.1f TF | x%.3f"%(k[0],k[1],a[k]["us"],a[k]["tflops"],b[k]["us"],b[k]["tflops"],a[k]["us"]/b[k]["us"]))
PY
fi
if has alltests; then
  echo "== pytest -m gpu (all)"
  timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  grep -E "PSNR|passed|failed|error" $OUT/pytest_gpu.log | tail -n 20
fi
if has bench; then
  echo "== bench"
  timeout 900 python bench.py --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
  grep -o '"value": [0-9.]*' $OUT/bench.log | head -1
  grep -o '"cpu_baseline": {[^}]*}' $OUT/bench.log | cut -c1-400
fi
if has rocprof; then
  echo "== rocprofv3 kernel stats of the bench command"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?" )
  find $OUT/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
fi
if has pmc; then
  echo "== PMC: HBM traffic of the eager denoising step, paired with the traced wrapper calls"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && ANIP_CALL_TRACE=$OUT/pmc_calls.json timeout 600 rocprofv3 --pmc $CTR --kernel-trace -f csv -d $OUT/pmc_step/$CTR -o p -- python $GRAFT_REPO_ROOT/tools/pmc_unet_step.py 2 > $OUT/pmc_step_$CTR.log 2>&1; echo "pmc $CTR rc=$?" )
  done
  find $OUT/pmc_step -name "*kernel_trace*" -delete 2>/dev/null
  python tools/pmc_summarize.py $OUT/pmc_step $OUT/pmc_step_summary.json --families --calls $OUT/pmc_calls.json 2>&1 | tail -n 2
  find $OUT/pmc_step -name "*counter_collection*" -size +6M -delete 2>/dev/null
fi
du -sh $OUT
