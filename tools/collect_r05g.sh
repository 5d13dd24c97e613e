#!/bin/bash
# Round 5: ragged (SAM-like) workloads on the shipped kernels + the sequence tests
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_sequence.py -m gpu -q -s 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2500 | grep -v "^$\|^   per-frame\|^   frame\|^   keyframe\|^   mapping" | tail -60) > $OUT/pytest_sequence.txt
for S in 64 300 1200; do
  timeout 900 python bench.py --shape blobs --segments $S --no-cpu-baseline --no-pmc > $OUT/bench_blobs_${S}.json 2> $OUT/bench_blobs_${S}.err
done
timeout 900 python bench.py --segments 128 --no-cpu-baseline > $OUT/bench_n1_seg128.json 2> $OUT/bench_seg128.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_blobs64 -o bench -- python $GRAFT_REPO_ROOT/bench.py --shape blobs --segments 64 --no-cpu-baseline --no-extras --no-pmc > $GRAFT_REPO_ROOT/$OUT/bench_blobs_64_under_rocprof.json 2>/dev/null)
tail -3 $OUT/pytest_sequence.txt | cut -c1-300
python - <<'PY'
import json
for f in ("bench_blobs_64", "bench_blobs_300", "bench_blobs_1200", "bench_n1_seg128"):
    try:
        d = json.load(open(f"gpurun_out/r05g/{f}.json"))
        print(f, round(d["value"]), round(d["roofline"]["frac"], 4), d.get("frame_pairs_per_sec"), d.get("frame_pairs_status", {}).get("flagged_failed"), d.get("frame_pairs_status", {}).get("silent_failures"), d.get("frame_pairs_per_sec_near_start_slot_batching"))
    except Exception as e:
        print(f, "no line", e)
PY
ls $OUT/stats_blobs64/* 2>/dev/null | head
