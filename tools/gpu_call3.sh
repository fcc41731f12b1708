#!/bin/bash
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for WHO in 1 2; do for K in 2 4 8; do
  ANIP_GEMM2_DBG=$((K*256 + WHO*65536)) timeout 200 python tools/bench_kernels.py --only=gemm > $OUT/mb_who${WHO}_k$K.jsonl 2>&1
done; done
timeout 200 python tools/bench_kernels.py --only=gemm > $OUT/mb_base.jsonl 2>&1
wc -l $OUT/*.jsonl
