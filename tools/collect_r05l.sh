#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05l
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_sigma05.py tests/test_gpu_fullsize.py -m gpu -q -s 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1200 | grep -v "^$" | tail -40) > $OUT/pytest.txt
(timeout 600 python tools/alone_probe.py all 2>&1 | grep shipped) > $OUT/alone_probe.txt
tail -6 $OUT/pytest.txt | cut -c1-300; cat $OUT/alone_probe.txt
