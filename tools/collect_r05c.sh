#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_sigma05.py tests/test_gpu_fullsize.py -m gpu -q -s 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1500 | grep -v "^$" | tail -150) > $OUT/pytest_pairs.txt
(timeout 600 python tools/verdict_sweep.py --variants shipped,no_retry --npz $OUT/verdict_sweep.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep.txt
(timeout 600 python tools/verdict_sweep.py --shape blobs --starts 3072 --alone "" --variants shipped,no_retry --npz $OUT/verdict_sweep_blobs.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_blobs.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -4 $OUT/pytest_pairs.txt; grep "==\|SILENT\|missed\|second\|silent\|false alarm\|flagged [0-9]" $OUT/verdict_sweep.txt $OUT/verdict_sweep_blobs.txt | cut -c1-420
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05c/bench_n1.json"))
print(d["value"], d["roofline"]["frac"], d.get("frame_pairs_per_sec"), d.get("frame_pairs_status"), d.get("timed_regions"))
PY
