#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05p
mkdir -p $OUT
if [ "$1" = "test" ]; then (timeout 900 python -m pytest tests/test_gpu_window_gn.py tests/test_gpu_sequence.py tests/test_gpu_drivers.py -m gpu -q -x 2>&1 | tail -15) > $OUT/pytest.txt; tail -5 $OUT/pytest.txt; fi
if [ "$1" = "prof" ]; then
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats_window -o w -- python $GRAFT_REPO_ROOT/tools/window_bench.py 2 > /dev/null 2>&1)
  head -12 $OUT/stats_window/*/w_kernel_stats.csv | cut -c1-200
  (timeout 600 python -m pytest tests/test_gpu_sequence.py -m gpu -q -s 2>&1 | grep -i "frames/s\|fps\|passed\|failed" | cut -c1-300)
fi
timeout 300 python tools/window_bench.py ${2:-2} 2>&1 | grep -v "^make\|amdgpu.ids" > $OUT/window_bench.txt
grep "gn:\|update kernel" $OUT/window_bench.txt | cut -c1-600
