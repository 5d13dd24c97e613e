#!/usr/bin/env python
"""Developer tool: per-kernel means of the counters in a rocprofv3 --pmc output directory.
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES ... --output-format csv -d DIR -o x -- <cmd>;  python tools/pmc_summary.py DIR [substr]"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_"
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> counter -> dispatch -> sum over the rows of the dispatch
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if sub not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            acc[short][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for k, c in sorted(acc.items()):
    n = max(len(v) for v in c.values())
    print(f"{k} ({n} dispatches): " + "  ".join(f"{cn} {sum(v.values()) / len(v):.4g}" for cn, v in sorted(c.items())))
