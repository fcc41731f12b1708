#!/bin/bash
# round-6 GPU session driver (one gpurun call = one invocation): `bash tools/gpu_round6.sh <tag> <step> [<step> ...]`, steps run in order:
#   env:VAR=V      export VAR=V for the following steps (library / engine switches)
#   ktests:<expr>  kernel tests  pytest tests/test_hip_ops.py -k "<expr>"
#   fbench:<n>     tools/exp_fused_blocks.py (round 5's fused kernels against the launches they replace) -> fbench_<n>.jsonl
#   kbench:<n> / abench:<n> / nbench:<n>   tools/bench_kernels.py gemm,conv / attn / norm -> *_<n>.jsonl
#   qbench:<n>     short whole-clip bench (6 clips, no CPU baseline / roofline) under the current environment: whole-clip A/B pairs
#   models / parity / alltests / smoke   model-level tests, fixture parity tests, the whole -m gpu suite, __graft_entry__.smoke()
#   bench / benchx / rocprof / pmc / pmck:<name>:<what>   as in tools/gpu_round4.sh
TAG=${1:-r06a}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# the snapshot carries no .git: the caller writes the commit the tree was taken at into .anip_commit (recorded by the PMC summary)
export ANIP_COMMIT=${ANIP_COMMIT:-$(cat .anip_commit 2>/dev/null)}
for STEP in "$@"; do
  echo "=================== $STEP"
  case $STEP in
  env:*) export "${STEP#env:}"; echo "exported ${STEP#env:}" ;;
  ktests:*)
    E=${STEP#ktests:}; N=$(echo "$E" | tr -c 'a-zA-Z0-9_' '_')
    timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "$E" > $OUT/ktests_$N.log 2>&1; echo "rc=$?" >> $OUT/ktests_$N.log
    grep -E "^FAILED|^ERROR|passed|failed|rc=|Error|error:|assert" $OUT/ktests_$N.log | tail -n 25 ;;
  fbench:*)
    N=${STEP#fbench:}
    timeout 600 python tools/exp_fused_blocks.py > $OUT/fbench_$N.jsonl 2>&1; echo "rc=$?"; tail -n 12 $OUT/fbench_$N.jsonl | cut -c1-220 ;;
  kbench:*)
    N=${STEP#kbench:}
    timeout 400 python tools/bench_kernels.py --only=gemm,conv $KBENCH_FLAGS > $OUT/kbench_$N.jsonl 2>&1; echo "rc=$?" ;;
  abench:*)
    N=${STEP#abench:}
    timeout 300 python tools/bench_kernels.py --only=attn > $OUT/abench_$N.jsonl 2>&1; echo "rc=$?"; tail -n 12 $OUT/abench_$N.jsonl | cut -c1-220 ;;
  nbench:*)
    N=${STEP#nbench:}
    timeout 300 python tools/bench_kernels.py --only=norm > $OUT/nbench_$N.jsonl 2>&1; echo "rc=$?"; tail -n 12 $OUT/nbench_$N.jsonl | cut -c1-220 ;;
  qbench:*)
    N=${STEP#qbench:}
    timeout 600 python bench.py --steps ${QSTEPS:-6} --warmup 2 --no-cpu-baseline --no-roofline --no-async-leg --no-extra-configs > $OUT/qbench_$N.log 2>&1; echo "rc=$?"
    grep -o '"value": [0-9.]*' $OUT/qbench_$N.log | head -1; grep -o '"ms_per_step": [0-9.]*' $OUT/qbench_$N.log | head -1
    grep -o '"per_clip_ms": \[[^]]*\]' $OUT/qbench_$N.log | head -1
    grep -E "Error|error|Traceback" $OUT/qbench_$N.log | head -5 ;;
  tbench:*)   # tbench:<name> — 2 clips with ANIP_PIPE_TIMING=1 (synchronised wall time per pipeline stage) under the current environment
    N=${STEP#tbench:}
    ANIP_PIPE_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-configs > $OUT/tbench_$N.log 2>&1; echo "rc=$?"
    grep -iE "refnet|unet|vae|pose|clip|total" $OUT/tbench_$N.log | tail -n 6 | cut -c1-400 ;;
  hbench:*)   # hbench:<name> — 10 clips with ANIP_PIPE_TIMING=host (host time per stage, no synchronize): where the host stalls
    N=${STEP#hbench:}
    ANIP_PIPE_TIMING=host timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-configs --no-async-leg > $OUT/hbench_$N.log 2>&1; echo "rc=$?"
    grep "pipe timing" $OUT/hbench_$N.log | tail -n 10 | cut -c1-700 ;;
  skbench:*)  # skbench:<name> — the tile-starved shapes under the current environment
    N=${STEP#skbench:}
    timeout 300 python tools/bench_kernels.py --only=sk > $OUT/skbench_$N.jsonl 2>&1; echo "rc=$?"
    python - <<PY
import json
for l in open("$OUT/skbench_$N.jsonl"):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if "us" in r: print("%-8s %-36s %8.1f us %7.1f TF"%(r["kernel"][:8], r["tag"][:36], r["us"], r["tflops"]))
PY
    ;;
  cputhreads) timeout 900 python tools/cpu_threads_probe.py 16 32 64 2>&1 | grep threads | tee $OUT/cpu_threads_probe.jsonl ;;
  gtests:*)   # gtests:<file>:<expr>  pytest tests/<file> -k "<expr>"
    AB=${STEP#gtests:}; Fi=${AB%%:*}; E=${AB#*:}; N=$(echo "$Fi$E" | tr -c 'a-zA-Z0-9_' '_')
    timeout 1500 python -m pytest tests/$Fi -m gpu -q -k "$E" > $OUT/gtests_$N.log 2>&1; echo "rc=$?" >> $OUT/gtests_$N.log
    grep -E "^FAILED|^ERROR|passed|failed|rc=|Error|error:|assert" $OUT/gtests_$N.log | tail -n 25 ;;
  cgroup)     # CPU quota of the box's container and its throttling counters (host-stall diagnosis)
    echo "nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
    python -c "import torch; print('torch threads', torch.get_num_threads(), torch.get_num_interop_threads())" ;;
  smoke)
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?"; tail -n 2 $OUT/smoke.log | cut -c1-300 ;;
  models)
    timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -q -x > $OUT/models.log 2>&1; echo "rc=$?" >> $OUT/models.log
    grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/models.log | tail -n 12 ;;
  parity)
    timeout 1500 python -m pytest tests/test_gpu_real_width.py -m gpu -q -s -k "fixture" > $OUT/parity.log 2>&1; echo "rc=$?" >> $OUT/parity.log
    grep -E "PSNR|^FAILED|^ERROR|passed|failed|rc=" $OUT/parity.log | cut -c1-400 | tail -n 16 ;;
  alltests)
    timeout 1800 python -m pytest tests -m gpu -q > $OUT/alltests.log 2>&1; echo "rc=$?" >> $OUT/alltests.log
    grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/alltests.log | tail -n 12 ;;
  bench)
    timeout 1500 python bench.py --steps 20 --warmup 5 --table-dir $OUT > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
    grep -o '"value": [0-9.]*' $OUT/bench.log | head -1; grep -o '"ms_per_step": [0-9.]*' $OUT/bench.log | head -1 ;;
  benchx)
    timeout 1500 python bench.py --extra-configs --table-dir $OUT > $OUT/benchx.log 2>&1; echo "bench rc=$?" | tee -a $OUT/benchx.log
    grep -o '"value": [0-9.]*' $OUT/benchx.log | head -1 ;;
  rocprof)
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs --no-async-leg > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?" )
    find $OUT/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null ;;
  pmc)
    for CTR in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && ANIP_CALL_TRACE=$OUT/pmc_calls.json timeout 600 rocprofv3 --pmc $CTR --kernel-trace -f csv -d $OUT/pmc_step/$CTR -o p -- python $GRAFT_REPO_ROOT/tools/pmc_unet_step.py 2 > $OUT/pmc_step_$CTR.log 2>&1; echo "pmc $CTR rc=$?" )
    done
    find $OUT/pmc_step -name "*kernel_trace*" -delete 2>/dev/null
    python tools/pmc_summarize.py $OUT/pmc_step $OUT/pmc_step_summary.json --families --calls $OUT/pmc_calls.json 2>&1 | tail -n 2
    find $OUT/pmc_step -name "*counter_collection*" -size +6M -delete 2>/dev/null ;;
  pmck:*)     # pmck:<name>:<what>  — SQ / TCC counter passes over tools/pmc_kernels.py <what> under the current environment
    AB=${STEP#pmck:}; N=${AB%%:*}; W=${AB##*:}
    i=0
    for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      ( cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -f csv -d $OUT/pmck_$N/pass$i -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W > $OUT/pmck_${N}_pass$i.log 2>&1; echo "pmck $N pass $i rc=$?" )
    done
    find $OUT/pmck_$N -name "*kernel_trace*" -delete 2>/dev/null
    python tools/pmc_summarize.py $OUT/pmck_$N $OUT/pmck_${N}_summary.json 2>&1 | tail -n 1
    python - <<PY
import json
d=json.load(open("$OUT/pmck_${N}_summary.json"))
for r in d["kernels"]:
    m=r["mean"]
    if not any(t in r["kernel"] for t in ("gemm2", "attn", "ffn", "temporal", "ln_", "gn_")): continue
    g=m.get("GRBM_GUI_ACTIVE",0)/8
    wc=max(1,m.get("SQ_WAVE_CYCLES",1))
    hit,miss=m.get("TCC_HIT_sum",0),m.get("TCC_MISS_sum",0)
    print("%-60s grid=%-8s n=%d cyc=%8.0f mfma=%.2f wait_any=%.2f wait_inst=%.2f active=%.2f valu=%.2f | L2 hit=%.3f req=%.2e | rd=%6.0fMB wr=%6.0fMB | lds_conf/idx=%.3f lds_wait=%.3f"%(
        r["kernel"][:60], r["grid"], r["launches"], g, r.get("mfma_busy_frac",-1), m.get("SQ_WAIT_ANY",0)/wc, m.get("SQ_WAIT_INST_ANY",0)/wc, m.get("SQ_ACTIVE_INST_ANY",0)/wc,
        m.get("SQ_ACTIVE_INST_VALU",0)/wc, hit/max(1,hit+miss), m.get("TCC_REQ_sum",0), r.get("hbm_read_bytes_per_launch",0)/1e6, r.get("hbm_write_bytes_per_launch",0)/1e6,
        m.get("SQ_LDS_BANK_CONFLICT",0)/max(1,m.get("SQ_LDS_IDX_ACTIVE",1)), m.get("SQ_WAIT_INST_LDS",0)/wc))
PY
    ;;
  usepmc)
    cp $OUT/pmc_step_summary.json profiles/pmc_traffic_latest.json && echo "profiles/pmc_traffic_latest.json <- $OUT/pmc_step_summary.json" ;;
  *) echo "unknown step $STEP" ;;
  esac
done
