#!/bin/bash
TAG=${1:-r02r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x > $OUT/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -n 3 $OUT/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_real_width.py tests/test_dropin_surface.py -m gpu -q -x -s -k "not c2_reduced and not windowed" > $OUT/pytest_models.log 2>&1; echo "pytest models rc=$?"; grep -E "PSNR|passed|failed" $OUT/pytest_models.log | tail -8
timeout 600 python bench.py --steps 4 --no-cpu-baseline --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?"
grep -o '"value": [0-9.]*, "unit"\|"unet3d_call_ms": [0-9.]*' $OUT/bench.log
