#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_sigma05.py -m gpu -q -s -k "ragged" 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-900 | grep -v "^$" | tail -12) > $OUT/pytest.txt
cat $OUT/pytest.txt
