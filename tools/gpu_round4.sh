#!/bin/bash
# round-4 GPU session driver (one gpurun call = one invocation; a call costs ~20 s of box time beyond what its steps run): `bash tools/gpu_round4.sh <tag> <step> [<step> ...]`,
# steps run in the order given:
#   parity       fixture-based real-width parity tests (C2 4 steps / C5 768 / L=40 windows)
#   bisect       per-block error table hip vs fp32 oracle at 32x32 / 64x64 latents (tests/bisect_parity.py)
#   bench        the headline bench line (+ per-shape table);  bench2: a second run (cpu_baseline reproducibility)
#   rocprof      rocprofv3 --kernel-trace --stats of the bench command
#   pmc          HBM-traffic PMC passes over the eager denoising step, paired with the traced wrapper calls
#   alltests     the whole -m gpu suite
#   env:VAR=V    export VAR=V for the following steps (library switches, KBENCH_FLAGS=--cold, ANIP_LIB=<experiment build>)
#   kbench:<n> / nbench:<n> / ktests:<n> / kcmp:<a>:<b>   micro-benchmarks and kernel tests under the current environment,
#                A/B table of two kbench runs of this call
#   valurates / storepattern / ldsdma   instruction issue-rate, store-shape and LDS-DMA micro-benchmarks (tools/exp_*.py)
#   qbench:<n>   short whole-clip bench (6 clips, no CPU baseline / roofline) under the current environment: whole-clip A/B pairs
#   mbench:<n>   the small once-per-step / once-per-clip kernels (tools/bench_kernels.py --only=misc)
#   abench:<n> / atests / pmck:<name>:<what>   attention micro-benchmark / tests, SQ + TCC counter passes over tools/pmc_kernels.py <what>
#   usepmc       make this call's PMC summary the profiles/pmc_traffic_latest.json that the following bench step reads
TAG=${1:-r04a}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for STEP in "$@"; do
  echo "=================== $STEP"
  case $STEP in
  env:*)      # env:VAR=VALUE — exported for the following steps
    export "${STEP#env:}"; echo "exported ${STEP#env:}" ;;
  nbench:*)   # nbench:<name> — tools/bench_kernels.py norm family under the current environment -> nbench_<name>.jsonl
    N=${STEP#nbench:}
    timeout 300 python tools/bench_kernels.py --only=norm > $OUT/nbench_$N.jsonl 2>&1; echo "rc=$?"
    python - <<PY
import json
for l in open("$OUT/nbench_$N.jsonl"):
    try: r=json.loads(l)
    except Exception: continue
    if "tag" in r: print("%-20s %-16s %8.1f us %7.0f GB/s"%(r["kernel"],r["tag"],r.get("ms",0)*1e3,r.get("gbps",0)))
PY
    ;;
  ntests)
    timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "groupnorm or layernorm" > $OUT/ntests.log 2>&1; echo "rc=$?" >> $OUT/ntests.log
    grep -E "^FAILED|passed|failed|rc=" $OUT/ntests.log | tail -n 8 ;;
  kbench:*)   # kbench:<name> — tools/bench_kernels.py gemm,conv under the current environment -> kbench_<name>.jsonl
    N=${STEP#kbench:}
    timeout 400 python tools/bench_kernels.py --only=gemm,conv $KBENCH_FLAGS > $OUT/kbench_$N.jsonl 2>&1; echo "rc=$?" ;;   # env:KBENCH_FLAGS=--cold
  kcmp:*)     # kcmp:<a>:<b>
    AB=${STEP#kcmp:}; A=${AB%%:*}; B=${AB##*:}
    python - <<PY
import json
def load(p):
    d={}
    try:
        for l in open(p):
            try: r=json.loads(l)
            except Exception: continue
            if "tag" in r: d[(r["kernel"],r["tag"])]=r
    except FileNotFoundError: pass
    return d
a,b=load("$OUT/kbench_$A.jsonl"),load("$OUT/kbench_$B.jsonl")
for k in a:
    if k in b: print("%-8s %-38s $A %8.1f us %7.1f TF | $B %8.1f us %7.1f TF | x%.3f"%(k[0],k[1],a[k]["us"],a[k]["tflops"],b[k]["us"],b[k]["tflops"],a[k]["us"]/b[k]["us"]))
PY
    ;;
  ktests:*)   # ktests:<name> — GEMM / conv kernel tests under the current environment
    N=${STEP#ktests:}
    timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "gemm or conv3x3 or ffn" > $OUT/ktests_$N.log 2>&1; echo "rc=$?" >> $OUT/ktests_$N.log
    grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/ktests_$N.log | tail -n 25 ;;
  pmclist)
    rocprofv3 -L > $OUT/rocprofv3_counters.txt 2>&1; grep -c . $OUT/rocprofv3_counters.txt ;;
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
    if "gemm2" not in r["kernel"] and "attn" not in r["kernel"]: continue
    g=m.get("GRBM_GUI_ACTIVE",0)/8
    wc=max(1,m.get("SQ_WAVE_CYCLES",1))
    hit,miss=m.get("TCC_HIT_sum",0),m.get("TCC_MISS_sum",0)
    print("%-60s grid=%-8s n=%d cyc=%8.0f mfma=%.2f wait_any=%.2f wait_inst=%.2f active=%.2f | L2 hit=%.3f req=%.2e | rd=%6.0fMB wr=%6.0fMB | lds_conf/idx=%.3f"%(
        r["kernel"][:60], r["grid"], r["launches"], g, r.get("mfma_busy_frac",-1), m.get("SQ_WAIT_ANY",0)/wc, m.get("SQ_WAIT_INST_ANY",0)/wc, m.get("SQ_ACTIVE_INST_ANY",0)/wc,
        hit/max(1,hit+miss), m.get("TCC_REQ_sum",0), r.get("hbm_read_bytes_per_launch",0)/1e6, r.get("hbm_write_bytes_per_launch",0)/1e6,
        m.get("SQ_LDS_BANK_CONFLICT",0)/max(1,m.get("SQ_LDS_IDX_ACTIVE",1))))
PY
    ;;
  vaebatch)
    timeout 600 python tests/bisect_parity.py --net vaebatch --sizes 32 64 96 --frames 16 --backends hip --out $OUT/vae_batch_vs_single.json > $OUT/vae_batch_vs_single.log 2>&1; echo "rc=$?" >> $OUT/vae_batch_vs_single.log
    grep -E "VAE batch|rc=|Error" $OUT/vae_batch_vs_single.log | tail -n 8
    python - <<PY
import json
for r in json.load(open("$OUT/vae_batch_vs_single.json")):
    rows=list(r["backends"].values())[0]["rows"]
    print("h=%d:"%r["h"], [(k.replace("decoder.",""), "%.1e"%v["rel_max"]) for k,v in rows.items() if v["rel_max"]>0][:6])
PY
    ;;
  bisectvae)
    timeout 900 python tests/bisect_parity.py --net vae --sizes 16 32 64 96 --frames 1 --backends hip --out $OUT/bisect_vae_hip.json > $OUT/bisect_vae_hip.log 2>&1; echo "rc=$?" >> $OUT/bisect_vae_hip.log
    grep -E "VAE h|rc=" $OUT/bisect_vae_hip.log | tail -n 8 ;;
  parity)
    timeout 900 python -m pytest tests/test_gpu_real_width.py -m gpu -q -s -k "fixture" > $OUT/parity_fixtures.log 2>&1; echo "rc=$?" >> $OUT/parity_fixtures.log
    grep -E "PSNR|passed|failed|rror|rc=" $OUT/parity_fixtures.log | tail -n 12 ;;
  bisect)
    timeout 900 python tests/bisect_parity.py --sizes 32 64 --frames 2 --backends hip --out $OUT/bisect_hip.json > $OUT/bisect_hip.log 2>&1; echo "rc=$?" >> $OUT/bisect_hip.log
    grep -E "conv_out rel|rc=" $OUT/bisect_hip.log | tail -n 6 ;;
  alltests)
    timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
    grep -E "PSNR|passed|failed|error" $OUT/pytest_gpu.log | tail -n 24 ;;
  bench|bench2)
    timeout 900 python bench.py --table-dir $OUT > $OUT/$STEP.log 2>&1; echo "bench rc=$?" | tee -a $OUT/$STEP.log
    grep -o '"value": [0-9.]*' $OUT/$STEP.log | head -1
    grep -o '"cpu_baseline": {[^}]*}' $OUT/$STEP.log | cut -c1-300 ;;
  qbench:*)   # qbench:<name> — short bench (no cpu baseline, no roofline) under the current environment
    N=${STEP#qbench:}
    timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/qbench_$N.log 2>&1; echo "rc=$?"
    grep -o '"value": [0-9.]*' $OUT/qbench_$N.log | head -1; grep -o '"ms_per_step": [0-9.]*' $OUT/qbench_$N.log | head -1 ;;
  mbench:*)   # mbench:<name> — the small once-per-step / once-per-clip kernels (tools/bench_kernels.py --only=misc)
    N=${STEP#mbench:}
    timeout 300 python tools/bench_kernels.py --only=misc > $OUT/mbench_$N.jsonl 2>&1; echo "rc=$?"; tail -n 8 $OUT/mbench_$N.jsonl | cut -c1-200 ;;
  benchx)
    timeout 1200 python bench.py --extra-configs --table-dir $OUT > $OUT/benchx.log 2>&1; echo "bench rc=$?" | tee -a $OUT/benchx.log
    grep -o '"value": [0-9.]*' $OUT/benchx.log | head -1 ;;
  rocprof)
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/rocprof_bench.log 2>&1; echo "rocprof rc=$?" )
    find $OUT/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null ;;
  pmc)
    for CTR in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && ANIP_CALL_TRACE=$OUT/pmc_calls.json timeout 600 rocprofv3 --pmc $CTR --kernel-trace -f csv -d $OUT/pmc_step/$CTR -o p -- python $GRAFT_REPO_ROOT/tools/pmc_unet_step.py 2 > $OUT/pmc_step_$CTR.log 2>&1; echo "pmc $CTR rc=$?" )
    done
    find $OUT/pmc_step -name "*kernel_trace*" -delete 2>/dev/null
    python tools/pmc_summarize.py $OUT/pmc_step $OUT/pmc_step_summary.json --families --calls $OUT/pmc_calls.json 2>&1 | tail -n 2
    find $OUT/pmc_step -name "*counter_collection*" -size +6M -delete 2>/dev/null ;;
  abench:*)   # abench:<name> — reference / temporal attention micro-benchmark under the current environment
    N=${STEP#abench:}
    timeout 300 python tools/bench_kernels.py --only=attn > $OUT/abench_$N.jsonl 2>&1; echo "rc=$?"; cat $OUT/abench_$N.jsonl | tail -n 12 ;;
  atests)
    timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "attention or attn" > $OUT/atests.log 2>&1; echo "rc=$?" >> $OUT/atests.log
    grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/atests.log | tail -n 25 ;;
  valurates)  # issue rates of the VALU / transcendental / MFMA instructions the kernel models use (tools/exp_valu_rates.py)
    timeout 240 python tools/exp_valu_rates.py > $OUT/valu_rates.jsonl 2>&1; echo "rc=$?"; cat $OUT/valu_rates.jsonl | tail -n 30 ;;
  ldsdma)     # global -> LDS throughput of one CU vs access shape / queue depth (tools/exp_lds_dma.py)
    timeout 200 python tools/exp_lds_dma.py > $OUT/lds_dma.jsonl 2>&1; echo "rc=$?"; tail -n 12 $OUT/lds_dma.jsonl ;;
  storepattern)
    timeout 120 python tools/exp_store_pattern.py > $OUT/store_pattern.jsonl 2>&1; echo "rc=$?"; tail -n 8 $OUT/store_pattern.jsonl ;;
  usepmc)     # make this call's PMC summary the one bench.py reads (the committed copy is refreshed from it afterwards)
    cp $OUT/pmc_step_summary.json profiles/pmc_traffic_latest.json && echo "profiles/pmc_traffic_latest.json <- $OUT/pmc_step_summary.json" ;;
  *) echo "unknown step $STEP" ;;
  esac
done
du -sh $OUT
