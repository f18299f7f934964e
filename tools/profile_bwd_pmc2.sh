#!/bin/bash
# second PMC look at the cost-volume backward: LDS pipe, issue stalls.  tools/profile_bwd_pmc2.sh <lib.so>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
lib=$1; tag=$(basename $lib .so)
OUT=$REPO/gpurun_out/pmc2_bwd_$tag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA_WRREQ_sum TCC_ATOMIC_sum" ; do
  name=$(echo $set | tr ' ' '+' | cut -c1-50)
  SMVS_LIB_PATH=$lib rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- python $REPO/tools/bench_bwd.py > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
cat "$OUT/errors.log" 2>/dev/null
python $REPO/tools/rocpd_summary.py "$OUT" | grep "bwd_kernel" | sed 's/.*smvs::CostVolBwdP[a-z]* *//' | awk '{printf "%-28s %s\n", $1, $NF}' | sort -u
