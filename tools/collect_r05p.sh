#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05p
mkdir -p $OUT
timeout 600 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_ceiling.json 2> $OUT/bench_ceiling.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05p/bench_ceiling.json"))
print(round(d["value"]), d["roofline"]["frac"], d["roofline"].get("valu_ceiling"))
PY
