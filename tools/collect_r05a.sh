#!/bin/bash
# Round 5, first GPU call: the new verdict / second attempt / ragged slot queue under test, the 9216-start sweep, the bench line.
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_sigma05.py -m gpu -q -s 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1200 | tail -120) > $OUT/pytest_pairs.txt
(timeout 600 python tools/verdict_sweep.py --npz $OUT/verdict_sweep.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 900 python bench.py --shape blobs --segments 64 --no-cpu-baseline --no-pmc > $OUT/bench_blobs_64.json 2> $OUT/bench_blobs_64.err
tail -5 $OUT/pytest_pairs.txt; grep "==\|SILENT\|missed\|second" $OUT/verdict_sweep.txt | cut -c1-300; tail -3 $OUT/bench_n1.err; python - <<'PY'
import json
for f in ("bench_n1", "bench_blobs_64"):
    try:
        d = json.load(open(f"gpurun_out/r05a/{f}.json"))
        print(f, d["value"], d["roofline"]["frac"], d.get("frame_pairs_per_sec"), d.get("frame_pairs_status"), d.get("timed_regions"))
    except Exception as e:
        print(f, "no line", e)
PY
