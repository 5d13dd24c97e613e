#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05k
mkdir -p $OUT
V=d16c12,d16c16,d31c12,d31c16,d16c12_retry_d31c16,d16c12_retry_d8c16,d16c16_retry_d31c16
(timeout 900 python tools/verdict_sweep.py --shape blobs --starts 12288 --alone "" --variants $V 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_blobs.txt
(timeout 600 python tools/verdict_sweep.py --starts 4608 --alone "" --variants d16c12,d16c16,d31c12,d31c16 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep.txt
grep "==\|missed\|second" $OUT/verdict_sweep_blobs.txt $OUT/verdict_sweep.txt | cut -c1-230
