#!/bin/bash
# One GPU-box round: parity tests, the bench line, the rocprofv3 kernel summary of the same command, and
# the kernel micro-benchmarks.  Everything lands under gpurun_out/ (copied to profiles/ afterwards).
# usage (from the repo root, on the GPU box):  bash tools/gpu_round.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; 
timeout 900 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 3 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
tail -n 2 $OUT/bench.log
cp gpurun_out/bench_kernels_table.json $OUT/ 2>/dev/null
echo "== rocprofv3 kernel stats of the bench command"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?" )
find $OUT/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
find $OUT/prof -name "*stats*" | head
echo "== kernel microbench"
timeout 600 python tools/bench_kernels.py > $OUT/bench_kernels.jsonl 2>&1; echo "microbench rc=$?"
tail -n 5 $OUT/bench_kernels.jsonl
