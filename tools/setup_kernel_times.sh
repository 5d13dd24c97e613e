#!/bin/bash
# Developer tool (GPU box): per-kernel durations of the set-up passes (rocprofv3 --stats over tools/setup_bench.py):  [SP_BOXES=1] bash tools/setup_kernel_times.sh [pairs]
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/skt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/skt -o s -- python $GRAFT_REPO_ROOT/tools/setup_bench.py ${1:-384} > /tmp/skt.log 2>&1
grep granule /tmp/skt.log | cut -c1-220
python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/skt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_prep' in r['Name']: print(r['Name'].replace('(anonymous namespace)::','')[:60], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs'])/1e3), 'min us %.1f' % (float(r['MinNs'])/1e3))
PY
