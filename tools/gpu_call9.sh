#!/bin/bash
TAG=${1:-r02h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "groupnorm" > $OUT/pytest_gn.log 2>&1; echo "pytest gn rc=$?"; tail -n 3 $OUT/pytest_gn.log
echo slab; timeout 120 python tools/bench_kernels.py --only=norm 2>&1 | grep groupnorm
echo two-pass; ANIP_GN_SLAB_KB=0 timeout 120 python tools/bench_kernels.py --only=norm 2>&1 | grep groupnorm
