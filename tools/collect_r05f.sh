#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05f
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_sigma05.py tests/test_gpu_fullsize.py tests/test_gpu_sequence.py -m gpu -q -s 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2500 | grep -v "^$\|^   per-frame\|^   frame\|^   keyframe" | tail -150) > $OUT/pytest.txt
(timeout 600 python tools/verdict_sweep.py --variants shipped,no_retry,undamped --npz $OUT/verdict_sweep.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep.txt
(timeout 600 python tools/verdict_sweep.py --slots 768 --variants shipped,undamped --alone "" 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_768.txt
(timeout 600 python tools/verdict_sweep.py --shape blobs --starts 3072 --alone "" --variants shipped,no_retry,undamped --npz $OUT/verdict_sweep_blobs.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_blobs.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -4 $OUT/pytest.txt; grep "==\|SILENT\|missed\|second\|flagged [0-9]" $OUT/verdict_sweep.txt $OUT/verdict_sweep_768.txt $OUT/verdict_sweep_blobs.txt | cut -c1-260
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05f/bench_n1.json"))
print(d["value"], d["roofline"]["frac"], d.get("frame_pairs_per_sec"), d.get("frame_pairs_status"))
print(d["reference_start"]["slot_level_continuous_batching"]["frame_pairs_per_sec_with_M_slots"])
PY
