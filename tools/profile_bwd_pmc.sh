#!/bin/bash
# PMC look at the cost-volume backward (tools/bench_bwd.py): tools/profile_bwd_pmc.sh <lib.so>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
lib=$1; tag=$(basename $lib .so)
OUT=$REPO/gpurun_out/pmc_bwd_$tag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-50)
  SMVS_LIB_PATH=$lib rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc -- python $REPO/tools/bench_bwd.py > "$OUT/pmc_$name.log" 2>&1 || echo "failed: $set" >> "$OUT/errors.log"
done
python $REPO/tools/rocpd_summary.py "$OUT" | grep -v rocclr | grep "bwd_kernel" | cut -c1-400
