#!/bin/bash
# usage (through gpurun): tools/r6_chain4.sh <variantA> <variantB>: chain tests with the product library, stamps of the TUNING build, kernel-trace A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6
exec > gpurun_out/r6/chain4.txt 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k chain 2>&1 | tail -2
true
exec >> gpurun_out/r6/chain4.txt 2>&1
bash tools/ab_kt.sh $1 $2 k_conv_chain; bash tools/ab_kt.sh $1 $2 k_conv_chain
