#!/bin/bash
TAG=${1:-r02s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x > $OUT/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -n 2 $OUT/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_real_width.py tests/test_dropin_surface.py -m gpu -q -x -s -k "not c2_reduced" > $OUT/pytest_models.log 2>&1; echo "pytest models rc=$?"; grep -E "PSNR|passed|failed" $OUT/pytest_models.log | tail -10
timeout 200 python tools/bench_kernels.py --only=attn 2>&1 | grep ref_att | head -2 | cut -c1-160
