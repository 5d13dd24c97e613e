#!/bin/bash
# Round-3 evidence on a GPU box:  tools/collect_r03.sh   (outputs under gpurun_out/r03/, copied into profiles/ by hand)
export TMPDIR=/tmp
OUT=gpurun_out/r03
mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -12) > $OUT/pytest.txt
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-extras --no-pmc > $OUT/bench_under_rocprof.json 2>/dev/null
for N in 64 300 1200; do
  timeout 400 python bench.py --shape blobs --segments $N --no-cpu-baseline --sigma05-scenes 0 > $OUT/bench_blobs_$N.json 2> $OUT/bench_blobs_$N.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_blobs_$N -o bench -- python bench.py --shape blobs --segments $N --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1
done
timeout 400 python tools/run_configs.py 2>/dev/null | grep config > $OUT/configs.txt
SP_STREAM_TRACE= timeout 300 python tools/stream_bench.py 2>&1 | grep batches > $OUT/stream_bench.txt
timeout 200 python tools/setup_profile.py 128 2>/dev/null | grep "PairBatch of\|run_scheduled\|build:" > $OUT/setup.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_setup -o setup -- python tools/setup_profile.py 128 > /dev/null 2>&1
timeout 200 python tools/kbench.py --pairs 384 --tile-points 8192 --modes 1,16,1,16,1,16,0 --reps 40 2>/dev/null | grep level > $OUT/kbench_pixless.txt
find $OUT -name "*.csv" | head -30
tail -5 $OUT/pytest.txt; cat $OUT/configs.txt $OUT/stream_bench.txt $OUT/setup.txt $OUT/kbench_pixless.txt; tail -2 $OUT/bench_n1.err
