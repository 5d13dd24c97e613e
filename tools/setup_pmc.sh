#!/bin/bash
# Developer tool (GPU box): counters of the set-up passes (k_prep_*) over tools/setup_bench.py, one counter group per rocprofv3 pass.
#   bash tools/setup_pmc.sh <out-file> [pairs]
OUT=$GRAFT_REPO_ROOT/$1; G=${2:-384}
mkdir -p $(dirname $OUT); : > $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/setup_bench.py $G"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/spmc_$i -o x -- $CMD > /tmp/spmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/spmc_$i k_prep >> $OUT 2>&1 || tail -3 /tmp/spmc_$i.log >> $OUT
  echo >> $OUT
done
grep granule /tmp/spmc_1.log >> $OUT
