#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05m
mkdir -p $OUT
(timeout 900 python tools/hard_ragged_probe.py 2437 9847 2>&1 | grep -v "^make\|amdgpu.ids") > $OUT/hard_ragged_probe2.txt
cat $OUT/hard_ragged_probe2.txt | cut -c1-250
