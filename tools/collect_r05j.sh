#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-400) > $OUT/smoke.txt
(timeout 900 python -m pytest tests/test_gpu_window_gn.py tests/test_gpu_window.py tests/test_gpu_drivers.py -m gpu -q 2>&1 | tail -4) > $OUT/pytest.txt
cat $OUT/smoke.txt; cat $OUT/pytest.txt
