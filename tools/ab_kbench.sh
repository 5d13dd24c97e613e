#!/bin/bash
# Developer tool: A/B the cost kernel across variant libraries on ONE box, interleaved (ABAB) so that clock drift hits all alike.
#   tools/ab_kbench.sh "base unroll128 ..." [kbench args]
VARS=$1; shift
for rep in 1 2 3; do
  for v in $VARS; do
    echo -n "$v: "
    SP_HIP_LIB=$PWD/super_primitive_amd/csrc/variants/libsp_$v.so python tools/kbench.py --pairs 384 --tile-points 8192 --modes 1 --reps 60 "$@" 2>/dev/null | grep level | sed 's/.*median/median/; s/alg.*//'
  done
done
