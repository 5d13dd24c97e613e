#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
V=undamped,d8c12,d8c8,d8c16,d16c12,d4c12,d8c25_tol1e-2,d8c12_direct_L1,d8c12_eps1e-2
(timeout 600 python tools/verdict_sweep.py --shape blobs --starts 3072 --alone "" --variants $V 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_blobs.txt
(timeout 600 python tools/verdict_sweep.py --starts 4608 --alone "" --variants $V 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep.txt
(timeout 600 python tools/verdict_sweep.py --starts 3072 --slots 768 --alone "" --variants undamped,d8c12,d8c12_direct_L1 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_768.txt
