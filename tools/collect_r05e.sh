#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
V=shipped,undamped,d8,d4,d16,d8_0.25,d4_0.5,d8_1,d8_then_L2,d8_cap12_then_L2,undamped_retry_d8_1
(timeout 600 python tools/verdict_sweep.py --shape blobs --starts 3072 --alone "" --variants $V --npz $OUT/verdict_sweep_blobs.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_blobs.txt
(timeout 600 python tools/verdict_sweep.py --starts 4608 --alone "" --variants $V 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep.txt
