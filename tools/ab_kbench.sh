#!/bin/bash
# Developer tool: A/B the cost kernel across variant libraries on ONE box, interleaved (ABAB) so that clock drift hits all alike.
#   tools/ab_kbench.sh "base unroll128 ..." [mode] [more kbench args]      (mode: 1 = GN pass (default), 0 = gradient pass)
VARS=$1; MODE=${2:-1}; shift; shift
for rep in 1 2 3; do
  for v in $VARS; do
    echo -n "$v: "
    SP_HIP_LIB=$PWD/super_primitive_amd/csrc/variants/libsp_$v.so python tools/kbench.py --pairs 384 --tile-points 8192 --modes $MODE --reps 60 "$@" 2>/dev/null | grep level | sed 's/.*median/median/; s/alg.*//'
  done
done
