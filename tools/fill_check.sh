#!/bin/bash
# Developer tool (GPU box): parity of the batched preparation + per-kernel times of the set-up passes (128 keyframes)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
python -m pytest tests/test_gpu_pairs.py -m gpu -q -x -k "prepar or raw or stream or table" 2>&1 | tail -3 > $OUT/t.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fs -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > $OUT/setup.log 2>&1
python - <<'PY' > $OUT/kernels.txt
import csv, glob
for f in glob.glob('/tmp/fs/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_prep' in r['Name']: print(r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'])
PY
if [ -n "$2" ]; then
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/fpA -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/fpB -o x -- python $GRAFT_REPO_ROOT/tools/setup_profile.py 128 > /dev/null 2>&1
(python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/fpA k_prep_fill; python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/fpB k_prep_fill) > $OUT/pmc.txt 2>&1
fi
cat $OUT/t.log $OUT/kernels.txt; cat $OUT/pmc.txt 2>/dev/null | cut -c1-500; grep "build:\|per pair" $OUT/setup.log
