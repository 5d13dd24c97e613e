#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05h
mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^make\|amdgpu.ids" | tail -15) > $OUT/pytest.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300) > $OUT/smoke.txt
tail -6 $OUT/pytest.txt; cat $OUT/smoke.txt
