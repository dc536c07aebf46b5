#!/bin/bash
# usage (through gpurun): tools/ab_libs.sh <variantA> <variantB> [rounds] -- <python command reading SSDN_HIP_LIB>
# runs the command alternately with the two library builds (tools/build_variant.sh) on the same box
cd ${GRAFT_REPO_ROOT:-.}
A=$1; B=$2; shift 2
R=2
if [ "$1" != "--" ]; then R=$1; shift; fi
shift
for r in $(seq $R); do
  for v in $A $B; do
    echo "== $v (round $r)"
    SSDN_HIP_LIB=$PWD/tools/_variants/$v/libssdn_hip.so "$@" 2>&1 | grep -v amdgpu.ids
  done
done
