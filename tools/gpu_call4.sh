#!/bin/bash
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attention" > $OUT/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_attn.log
timeout 200 python tools/bench_kernels.py --only=attn > $OUT/mb_attn.jsonl 2>&1
ANIP_ATTN_QH=2 timeout 200 python tools/bench_kernels.py --only=attn > $OUT/mb_attn_qh2.jsonl 2>&1
cat $OUT/mb_attn.jsonl $OUT/mb_attn_qh2.jsonl | grep ref_att
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_real_width.py -m gpu -q -x -k "not c2_reduced and not windowed and not c1_in_full" > $OUT/pytest_models.log 2>&1; echo "pytest models rc=$?"; tail -n 4 $OUT/pytest_models.log
timeout 600 python bench.py --steps 2 --no-cpu-baseline --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?"
grep -o '"value": [0-9.]*' $OUT/bench.log | head -1
ANIP_FUSED_FFN=1 timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-roofline > $OUT/bench_fused_ffn.log 2>&1; echo "bench(fused) rc=$?"
grep -o '"value": [0-9.]*' $OUT/bench_fused_ffn.log | head -1
