#!/bin/bash
# Developer tool (GPU box): SQ / TA / TCP counters of k_cost_pairs on the headline batch through tools/kbench.py, one counter group per pass.
#   bash tools/kernel_pmc.sh <out-file> [modes]
OUT=$1; MODES=${2:-1}
mkdir -p $(dirname $OUT); : > $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/kbench.py --pairs 384 --tile-points 8192 --granule 64 --reps 20 --modes $MODES"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/kpmc_$i -o x -- $CMD > /tmp/kpmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/kpmc_$i k_cost_pairs >> $GRAFT_REPO_ROOT/$OUT 2>&1 || tail -3 /tmp/kpmc_$i.log >> $GRAFT_REPO_ROOT/$OUT
done
