#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05p
mkdir -p $OUT
if [ "$1" = "test" ]; then (timeout 900 python -m pytest tests/test_gpu_window_gn.py tests/test_gpu_sequence.py tests/test_gpu_drivers.py -m gpu -q -x 2>&1 | tail -15) > $OUT/pytest.txt; tail -5 $OUT/pytest.txt; fi
timeout 300 python tools/window_bench.py ${2:-2} 2>&1 | grep -v "^make\|amdgpu.ids" > $OUT/window_bench.txt
grep "gn:\|update kernel" $OUT/window_bench.txt | cut -c1-600
