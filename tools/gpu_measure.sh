#!/bin/bash
# round-end measurement on one MI355X: parity suite, the bench line (+C4/C5 extras), rocprofv3 kernel stats of the
# bench command, PMC passes (HBM traffic of the bench command without graphs; SQ / TCC counters on the kernel set).
# Everything lands under gpurun_out/<tag>/ and is copied into profiles/ afterwards.
TAG=${1:-r02m}
WHAT=${2:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [[ $WHAT == all || $WHAT == *tests* ]]; then
  echo "== pytest -m gpu"
  timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  grep -E "PSNR|passed|failed|error" $OUT/pytest_gpu.log | tail -n 20
fi
if [[ $WHAT == all || $WHAT == *bench* ]]; then
  echo "== bench (headline + extras)"
  timeout 1200 python bench.py --extra-configs --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
  grep -o '"value": [0-9.]*' $OUT/bench.log | head -1
fi
if [[ $WHAT == all || $WHAT == *rocprof* ]]; then
  echo "== rocprofv3 kernel stats of the bench command"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?" )
  find $OUT/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
  find $OUT/prof -name "*stats*" | head -3
fi
if [[ $WHAT == all || $WHAT == *pmc* ]]; then
  echo "== PMC: HBM traffic of the denoising step (eager UNet3D forward at the C2 shapes; see tools/pmc_unet_step.py)"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $CTR --kernel-trace -T -f csv -d $OUT/pmc_step/$CTR -o p -- python $GRAFT_REPO_ROOT/tools/pmc_unet_step.py 2 > $OUT/pmc_step_$CTR.log 2>&1; echo "pmc $CTR rc=$?" )
  done
  find $OUT/pmc_step -name "*kernel_trace*" -delete 2>/dev/null
  python tools/pmc_summarize.py $OUT/pmc_step $OUT/pmc_step_summary.json --families 2>&1 | tail -n 1
  find $OUT/pmc_step -name "*counter_collection*" -size +6M -delete 2>/dev/null
  echo "== PMC: SQ / TCC counters on the kernel set"
  i=0
  for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -T -f csv -d $OUT/pmc_kernels/pass$i -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py all > $OUT/pmc_kernels_pass$i.log 2>&1; echo "pmc kernels pass $i rc=$?" )
  done
  find $OUT/pmc_kernels -name "*kernel_trace*" -delete 2>/dev/null
  python tools/pmc_summarize.py $OUT/pmc_kernels $OUT/pmc_kernels_summary.json 2>&1 | tail -n 1
fi
du -sh $OUT
