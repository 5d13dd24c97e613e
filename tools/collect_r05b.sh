#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_sigma05.py -m gpu -q -s 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-1500 | grep -v "^$" | tail -150) > $OUT/pytest_pairs.txt
(timeout 600 python tools/verdict_sweep.py --npz $OUT/verdict_sweep.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep.txt
(timeout 600 python tools/verdict_sweep.py --shape blobs --starts 3072 --alone "" --npz $OUT/verdict_sweep_blobs.npz 2>&1 | grep -v "^make\|amdgpu.ids" | cut -c1-2000) > $OUT/verdict_sweep_blobs.txt
tail -4 $OUT/pytest_pairs.txt; grep "==\|SILENT\|missed\|second\|silent\|false alarm" $OUT/verdict_sweep.txt $OUT/verdict_sweep_blobs.txt | cut -c1-420
