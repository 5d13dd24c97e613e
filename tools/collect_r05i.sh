#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05i
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_window_gn.py tests/test_gpu_sequence.py -m gpu -q -s -k "mono_init or persistent" 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1800 | grep -v "^$" | tail -40) > $OUT/pytest.txt
/usr/bin/time -v timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -25 $OUT/pytest.txt | cut -c1-900; grep "Elapsed" $OUT/bench_n1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05i/bench_n1.json"))
print(d["value"], d["roofline"]["frac"], d.get("frame_pairs_per_sec"), d.get("frame_pairs_per_sec_ragged_masks"), d.get("frame_pairs_status"))
print({k: d["reference_start_ragged_masks"][k] for k in ("pairs", "frame_pairs_per_sec", "converged_fraction", "verdict")})
PY
