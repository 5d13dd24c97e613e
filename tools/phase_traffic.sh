#!/bin/bash
# Developer tool (GPU box): HBM-side bytes per cost-pass launch of every phase of the reference-start schedule (1536 pairs resident, all in one phase)
# next to the phase's algorithmic bytes:  bash tools/phase_traffic.sh <out-file>
OUT=$GRAFT_REPO_ROOT/$1; mkdir -p $(dirname $OUT); : > $OUT
cd /tmp; export TMPDIR=/tmp
for ph in "2 4" "1 2" "0 2" "0 1"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pt
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pt -o x -- python $GRAFT_REPO_ROOT/tools/phase_one.py $ph > /tmp/pt.log 2>&1
    [ $c = FETCH_SIZE ] && grep "^level" /tmp/pt.log >> $OUT
    python - $c >> $OUT <<'PY'
import csv, glob, collections, sys
c = sys.argv[1]
acc = collections.defaultdict(float); grid = {}
for path in glob.glob("/tmp/pt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_cost_pairs" not in r["Kernel_Name"]: continue
        acc[r["Dispatch_Id"]] += float(r["Counter_Value"]); grid[r["Dispatch_Id"]] = int(r["Grid_Size"])
by = collections.defaultdict(list)
for d, v in acc.items(): by[grid[d]].append(v)
g, v = max(by.items(), key=lambda kv: len(kv[1]))
print(f"    {c}: {len(v)} cost launches (grid {g} threads): mean {sum(v) / len(v) / 1e3:.1f} MB per launch as the counter reports it (KB units)")
PY
  done
done
