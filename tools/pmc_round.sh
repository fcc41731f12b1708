#!/bin/bash
# PMC counter passes (separate runs, --kernel-trace only) over tools/pmc_kernels.py; CSVs under gpurun_out/<tag>/pmc*
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -T -f csv -d $OUT/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py all > $OUT/pmc$i.log 2>&1
  echo "pass $i ($CTRS) rc=$?"
done
ls -la $OUT/pmc*/ | head -30
