#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05n
mkdir -p $OUT
V=d8c12,d12c12,d16c12,d20c12,d24c12,d31c12
(timeout 900 python tools/verdict_sweep.py --shape blobs --segments 300 --starts 3072 --alone "" --variants $V 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1200) > $OUT/sweep_blobs300.txt
(timeout 900 python tools/verdict_sweep.py --shape blobs --segments 1200 --starts 1536 --alone "" --variants $V 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1200) > $OUT/sweep_blobs1200.txt
(timeout 900 python tools/verdict_sweep.py --segments 128 --starts 3072 --alone "" --variants $V 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1200) > $OUT/sweep_grid128.txt
(timeout 900 python tools/verdict_sweep.py --shape blobs --starts 12288 --alone "" --variants d12c12,d20c12,d24c12 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1200) > $OUT/sweep_blobs64.txt
grep "==\|missed\|second" $OUT/sweep_blobs300.txt $OUT/sweep_blobs1200.txt $OUT/sweep_grid128.txt $OUT/sweep_blobs64.txt | cut -c1-200
