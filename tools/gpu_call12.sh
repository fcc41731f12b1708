#!/bin/bash
TAG=${1:-r02o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm or conv or ffn" > $OUT/pytest_ops.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_ops.log
timeout 300 python tools/bench_kernels.py --only=gemm,conv > $OUT/mb.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/'+'r02o'+'/mb.jsonl'):
    try: r=json.loads(l)
    except: continue
    if 'kernel' in r: print(f"{r['kernel'][:6]} {r['tag'][:44]:44s} {r['us']:8.1f} us {r.get('tflops',0):7.1f} TF/s")
PY
