#!/bin/bash
TAG=${1:-r02n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attention" > $OUT/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_attn.log
timeout 200 python tools/bench_kernels.py --only=attn,norm 2>&1 | grep -E "ref_att|temporal" | cut -c1-200
