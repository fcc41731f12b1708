#!/bin/bash
# round-2 GPU call 2: main-loop cleanup microbench, stagger experiment, fused-FFN end-to-end check, quick bench
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== quick parity of the touched kernels"
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm or conv or attention or f16_to_u8 or window" > $OUT/pytest_ops.log 2>&1; echo "pytest rc=$?"
tail -n 3 $OUT/pytest_ops.log
echo "== microbench (new lib)"
timeout 300 python tools/bench_kernels.py --only=gemm,conv > $OUT/mb_new.jsonl 2>&1
for K in 4 8 16; do
  ANIP_GEMM2_DBG=$((K*256)) timeout 200 python tools/bench_kernels.py --only=gemm > $OUT/mb_stagger$K.jsonl 2>&1
done
ANIP_GEMM2_CFG=2 timeout 200 python tools/bench_kernels.py --only=gemm > $OUT/mb_wide.jsonl 2>&1
wc -l $OUT/mb_*.jsonl
echo "== bench quick (default / fused FFN)"
timeout 600 python bench.py --steps 2 --no-cpu-baseline --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?"
grep -o '"value": [0-9.]*' $OUT/bench.log | head -1
ANIP_FUSED_FFN=1 timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-roofline > $OUT/bench_fused_ffn.log 2>&1; echo "bench(fused) rc=$?"
grep -o '"value": [0-9.]*' $OUT/bench_fused_ffn.log | head -1
du -sh $OUT
