#!/bin/bash
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "ffn" > $OUT/pytest_ffn.log 2>&1; echo "pytest ffn rc=$?"; tail -n 3 $OUT/pytest_ffn.log
echo "v2:"; timeout 120 python tools/exp_ffn.py 2>&1 | tail -1
echo "v1:"; ANIP_FFN_V=1 timeout 120 python tools/exp_ffn.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_real_width.py -m gpu -q -x -k "not c2_reduced and not windowed and not c1_in_full" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest.log
ANIP_PIPE_TIMING=1 timeout 600 python bench.py --steps 2 --no-cpu-baseline --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?"
grep -E "pipe timing" $OUT/bench.log | tail -2 | cut -c1-400
grep -o '"value": [0-9.]*' $OUT/bench.log | head -1
