#!/bin/bash
# round-2 GPU call 1: the parity suite (incl. real-width BASELINE configs), the bench line, kernel variants side by side
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "PSNR|passed|failed|error" $OUT/pytest_gpu.log | tail -n 25
echo "== bench"
timeout 900 python bench.py --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
tail -c 3000 $OUT/bench.log
echo "== kernel variants"
timeout 300 python tools/bench_kernels.py --only=gemm,conv,attn,norm > $OUT/mb_new.jsonl 2>&1
ANIP_GEMM2_DBG=8 ANIP_ATTN_QH=1 timeout 300 python tools/bench_kernels.py --only=gemm,conv,attn > $OUT/mb_r1.jsonl 2>&1
ANIP_GEMM2_CFG=2 timeout 200 python tools/bench_kernels.py --only=gemm > $OUT/mb_wide.jsonl 2>&1
ANIP_GEMM2_DBG=9 timeout 200 python tools/bench_kernels.py --only=gemm > $OUT/mb_noepi.jsonl 2>&1
wc -l $OUT/mb_*.jsonl
echo "== pmc (SQ) on the kernel set"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -T -f csv -d $GRAFT_REPO_ROOT/$OUT/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py all > $GRAFT_REPO_ROOT/$OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $OUT/pmc1 $OUT/pmc1_summary.json 2>&1 | tail -n 2
find $OUT -name "*kernel_trace*" -size +4M -delete 2>/dev/null
du -sh $OUT
