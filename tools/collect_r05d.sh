#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05d
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_pairs.py -m gpu -q -s -k "box or batched_prep or keyframe_record" 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1500 | grep -v "^$" | tail -60) > $OUT/pytest_boxes.txt
timeout 900 python bench.py --no-cpu-baseline --no-pmc > $OUT/bench_n1.json 2> $OUT/bench_n1.err
bash tools/ab_kbench.sh "base onercp base onercp" 1 --granule 64 > $OUT/ab_onercp.txt 2>&1; cat $OUT/ab_onercp.txt; tail -4 $OUT/pytest_boxes.txt; tail -3 $OUT/bench_n1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05d/bench_n1.json"))
print(d["value"], d["roofline"]["frac"], d.get("frame_pairs_per_sec"))
f = d["from_raw_frames"]
print("setup", f["setup_ms"], "opt", f["optimise_ms"], "with boxes", f.get("with_segment_boxes"))
print({k: (round(v["ms"], 3), round(v["frac"], 3)) for k, v in d["roofline_setup"]["passes"].items()})
PY
