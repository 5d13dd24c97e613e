cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o lp -- python $GRAFT_REPO_ROOT/tools/lone_pair_latency.py > /tmp/lp.log 2>&1
head -8 /tmp/lp.log | cut -c1-200
python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/lp/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['Percentage'])
PY
