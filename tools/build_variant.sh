#!/bin/bash
# Developer tool: build an A/B variant of libsp_hip.so from the working tree with a sed script applied to sp_cost.hip.
#   tools/build_variant.sh NAME 'sed-expression' [git-ref]   -> super_primitive_amd/csrc/variants/libsp_NAME.so
# (git-ignored, travels to the GPU box; select it with SP_HIP_LIB=...).  With a git ref the sources come from that commit.
set -e
NAME=$1; SED=$2; REF=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/spvariant_$NAME
rm -rf $W; mkdir -p $W/super_primitive_amd/csrc $W/include
if [ -n "$REF" ]; then
  for f in $(git -C $ROOT ls-tree --name-only $REF super_primitive_amd/csrc/ | grep -E "\.(hip|h)$|Makefile"); do git -C $ROOT show $REF:$f > $W/$f; done
  git -C $ROOT show $REF:include/sp_hip.h > $W/include/sp_hip.h
else
  cp $ROOT/super_primitive_amd/csrc/*.hip $ROOT/super_primitive_amd/csrc/*.h $ROOT/super_primitive_amd/csrc/Makefile $W/super_primitive_amd/csrc/
  cp $ROOT/include/sp_hip.h $W/include/
fi
[ -n "$SED" ] && sed -i "$SED" $W/super_primitive_amd/csrc/sp_cost.hip
make -C $W/super_primitive_amd/csrc -j8 > /dev/null
mkdir -p $ROOT/super_primitive_amd/csrc/variants
cp $W/super_primitive_amd/csrc/libsp_hip.so $ROOT/super_primitive_amd/csrc/variants/libsp_$NAME.so
echo built variants/libsp_$NAME.so
