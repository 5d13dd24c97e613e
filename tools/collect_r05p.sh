#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05p
mkdir -p $OUT
timeout 600 python tools/chain_profile.py 40 2>&1 | grep -v "^make\|amdgpu.ids" > $OUT/chain_profile.txt
head -3 $OUT/chain_profile.txt | cut -c1-400
