#!/bin/bash
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python tools/exp_attn_layout.py > $OUT/exp_attn_layout.jsonl 2>&1; cat $OUT/exp_attn_layout.jsonl | tail -6
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_gpu_models.py tests/test_gpu_real_width.py -m gpu -q -x -k "not c2_reduced and not windowed and not c1_in_full" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest.log
timeout 600 python bench.py --steps 2 --no-cpu-baseline --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?"
grep -o '"value": [0-9.]*' $OUT/bench.log | head -1
ANIP_FUSED_FFN=1 timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-roofline > $OUT/bench_fused_ffn.log 2>&1; echo "bench(fused) rc=$?"
grep -o '"value": [0-9.]*' $OUT/bench_fused_ffn.log | head -1
