#!/bin/bash
# Build the CURRENT csrc/ into tools/_variants/<name>/libssdn_hip.so without touching the product build (same flags; extra make
# arguments, e.g. TUNING=1, are passed on).  Select a variant with SSDN_HIP_LIB=tools/_variants/<name>/libssdn_hip.so (ssdn/hip/lib.py):
# how two states of a kernel are timed against each other on ONE box in one gpurun call (tools/ab_libs.sh).  The .so files are
# git-ignored but travel to the GPU box with the snapshot.
set -e
NAME=${1:?usage: build_variant.sh <name> [make args]}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/ssdn_variant_build_$NAME
mkdir -p $T/selfsupervised-denoising_amd/csrc $T/include $ROOT/tools/_variants/$NAME
cp -u $ROOT/selfsupervised-denoising_amd/csrc/*.hip $ROOT/selfsupervised-denoising_amd/csrc/*.h $ROOT/selfsupervised-denoising_amd/csrc/Makefile \
      $ROOT/selfsupervised-denoising_amd/csrc/*.py $ROOT/selfsupervised-denoising_amd/csrc/export.map $T/selfsupervised-denoising_amd/csrc/
cp -u $ROOT/include/*.h $T/include/
make -C $T/selfsupervised-denoising_amd/csrc -j8 LIB=$ROOT/tools/_variants/$NAME/libssdn_hip.so "$@" 2>&1 | tail -2
