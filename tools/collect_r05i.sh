#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05i
mkdir -p $OUT
S=$(date +%s); timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05i/bench_n1.json"))
print(d["value"], d["roofline"]["frac"], d.get("frame_pairs_per_sec"), d.get("frame_pairs_per_sec_ragged_masks"), d.get("frame_pairs_status"))
print({k: d["reference_start_ragged_masks"][k] for k in ("pairs", "frame_pairs_per_sec", "converged_fraction", "verdict")})
PY
